// net.h — policy/value net state shared by engine.cpp and net_kernels.hip
#pragma once
#include "common.h"
#include "../../include/omok_mi355x.h"

namespace omok {

constexpr int NET_TENSORS = 31;
constexpr int NC = 128; // RESIDUAL_CHANNELS (alpha-zero/src/network.rs:24)
constexpr int NM = 32;  // RESIDUAL_MIDDLE_CHANNELS (:25)
constexpr int NF = 512; // FC_0_SIZE / FC_1_SIZE (:29-30)
enum { FC0_AUTO = -1, FC0_FP6 = 0, FC0_F16 = 1, FC0_MIXED = 2 }; // MIXED: full rows as FC0_F16, the difference rows of sibling rounds as FC0_FP6
// fp6 correction terms are kept only while the probe's worst |dp|, |dv| stay below this.  Round 4: 3e-4 (was 5e-4): held-out positions exceeded the probe's figure by up to 1.7x
// (profiles/r03_precision_sweep.log) and the difference path of the search rounds adds up to 3e-4 against row-by-row evaluation, so 3e-4 x 1.7 + 3e-4 stays inside the 1e-3 contract
constexpr float NET_PROBE_LIMIT = 3e-4f;
// ... and the logits in front of the softmax / the value in front of tanh (north_star: "policy/value logits within 1e-3") within half of that bar
constexpr float NET_PROBE_LOGIT_LIMIT = 5e-4f;
// The limits above are the MARGIN the chooser keeps (a format is preferred only while inside them).  The CONTRACT is north_star's 1e-3: when even the most precise
// split-operand format (f16 correction terms) measures above it on the probe, no format of this family may be committed silently -- net_commit then falls back to the
// plain fp32 kernels (the arithmetic of the reference's AgentModel::evaluate_pv, agent_model.rs:116-134) and says so (Net::probe_outside, OMOK_STAT_PROBE_OUTSIDE).
constexpr float NET_PROBE_CONTRACT = 1e-3f;
constexpr int NET_PROBE_ROWS = 2048;

struct Net {
    int n = 0, hw = 0, rowp = 0, mode = 0;
    int cfg_mode = 0; // the mode the engine was created in (net_alloc); `mode` leaves it only for the fp32 fallback of a commit whose probe is outside the contract
    bool siblings = true; // search rounds may take the sibling path of the trunk / fc0 (false: OMOK_NET_F16X3_ROWS)
    int device = 0;   // HIP device ordinal (keys the per-device launch caches)
    int max_b = 0;
    bool committed = false;
    bool loaded[NET_TENSORS] = {};
    int64_t wsize[NET_TENSORS] = {};
    float* w[NET_TENSORS] = {}; // raw fp32 tensors in reference order (device)
    // outputs: p [max_b][ROWP] softmax probabilities (pad cells 0), v [max_b]
    float* p = nullptr;
    float* v = nullptr;
    float* vpre = nullptr; // [max_b] value head before tanh (omok_evaluate_logits)
    float* in_f32 = nullptr; // [max_b][3*HW] encoder.rs layout (evaluate_pv / step-wise API / f32 path)
    // ---- OMOK_NET_F32 scratch (chunked) ----
    int chunk = 0;
    float *sx = nullptr, *sh = nullptr, *sd = nullptr, *sg = nullptr, *s0 = nullptr, *s1 = nullptr;
    float *s0_x3 = nullptr, *s0_f32 = nullptr; // fp32 fallback of a split-precision engine: s0 is the logits buffer of the one mode and an fc scratch of the other
    int probe_outside = 0;    // verdict of the last commit's probe on the format it chose: 0 = inside the margin limits, 1 = outside them but inside the 1e-3 contract
                              // (committed, logged once), 2 = the f16 format itself is outside the contract: the engine evaluates with the fp32 kernels until the next commit
    // ---- OMOK_NET_F16X3: packed split-fp16 operands (see net_kernels.hip) ----
    void* wt_trunk = nullptr; // packed trunk weights (A-operand fragments, hi|lo)
    float* wt_first = nullptr; // conv_in weights/bias + all biases + depthwise taps, fp32
    void* wt_fc0 = nullptr;   // [kstep][ntile][hi|lo][lane][8] f16
    void* wt_fc1 = nullptr;
    void* wt_heads = nullptr;
    void* a_fc0 = nullptr;    // [max_b][KSTEPS][hi 16 | lo 16] f16 : trunk output = fc0 A operand
    void* h0 = nullptr;       // [max_b][512] hi|lo f16 : fc0 output = fc1 A operand
    size_t row_u4 = 0;        // a_fc0 row stride in uint4
    size_t part_rows = 0;     // capacity of the split-K partial slab `part` in rows (x 512 floats)
    float* part = nullptr;    // split-K fp32 partials of fc0 for small batches
    int32_t* d_chunk = nullptr; // [64][4] live row count of every row chunk of a forward (see forward_chunked)
    // sibling path of the trunk (N = 15): runs of sibling requests, remaining single rows, counters, per-workgroup base scratch
    void* d_groups = nullptr;
    int32_t* d_singles = nullptr;
    void* d_sib_rows = nullptr; // (request row, run) of the rows inside runs
    int32_t* d_gcnt = nullptr;  // [0] runs, [1] rows outside runs, [2] rows inside runs, [3] full rows (runs + singles), [4] fc0 window tiles,
                                // [8 + b] children whose window is bin b (SIB_CNT_INTS in all)
    // Round 6: the fc0 of a sibling round's FULL rows (the runs' bases evaluated this round + the single rows) depends on the base trunk only, not on the children kernel:
    // it is launched on a side stream behind the base trunk and joins the main stream in front of the window tiles, so that its workgroups take the CUs the persistent
    // children kernel's workgroups leave at their end (the children hold every CU's LDS: nothing else is resident before that).  side_on = false (the default, see
    // SIDE_STREAM_DEFAULT below): everything on the main stream.
    hipStream_t side = nullptr;
    hipEvent_t ev_base = nullptr, ev_full = nullptr;
    bool side_on = false;
    unsigned long long* d_work = nullptr; // [NET_WORK_COUNT] executed-work counters summed over the rounds since omok_reset_stats (k_group, k_bin_prefix add; only omok_get_stats reads)
    float* sib_h = nullptr;     // [run][3 blocks][225][32] the base passes' depthwise inputs
    // difference path: a child's fc0 input = its run's base row + a 7x7-window difference row
    uint32_t* d_sib_slot = nullptr;  // per row inside a run: window bin << 24 | rank inside the bin
    int32_t* d_bin_start = nullptr;  // [pixel] first slot of the children whose stone lands in that net pixel (a bin's pixels in row-major order, bins padded to whole 128-sample tiles)
    int32_t* d_tile_info = nullptr;  // per fc0 window tile: bin | live slots << 8 | the rectangle of window pixels its rows can differ in (y0 << 16 | y1 << 19 | x0 << 22 | x1 << 25)
    void* d_slot_desc = nullptr;     // per slot: (request row, full-row index of its base / of itself)
    void* d_rows = nullptr;          // [slot][2 q][49 window pixels] f16 parts, then residual parts: the difference rows
    size_t d_slots = 0;              // slots allocated
    // base cache: a leaf stays the expansion target of its tree for ~14 rounds, so its base evaluation (operand row, h grids, fp32 fc0 row)
    // is kept per game slot and reused while the tree's runs keep the same parent (~78 % of the runs of a configs[1] episode)
    int games = 0;                   // game slots of the engine (base slots [0, 2 games): two per game (SIB_WAYS); behind them: other runs of a round)
    size_t base_slots = 0;
    void* a_base = nullptr;          // [base_slots] operand rows of base positions (row_u4 each)
    float* facc = nullptr;           // [base_slots + max_b][512] fp32 fc0 rows: base slots, then the round's single rows
    int32_t* d_tags = nullptr;       // [games][2] (leaf node | slot << 16) of the bases in the game's slots (SIB_WAYS), most recently used first; -1 = none
    void* d_comp = nullptr;          // the round's positions to evaluate in full: (request row of the first child, base slot)
    bool gcnt_zeroed = false;        // the engine's k_scan of this round has zeroed d_gcnt (launch_trunk_siblings then skips k_zero_ints)
    bool fill_in_group = false;      // ... and left the dense request list to k_group (launch_scan(fill = false))
    int fill_side = 0, fill_k = 0;   // ... for this side's trees and this K: a forward that does not group after all writes the list itself (launch_fill)
    double children_launches[2] = {0.0, 0.0}; // sibling rounds by children kernel: [0] k_sib_children2, [1] k_sib_children
    bool sib_v2 = true;              // difference path: k_sib_children2 (one wave per child, growing windows) and the base-slot layout it reads
    bool win_rects = true;           // fc0 window tiles walk only the rectangle of window pixels their rows can differ in (k_bin_prefix); false (omok_debug_set_window_rects): the whole 7x7 window -- same bits
    bool base_cache = true;          // false (omok_debug_set_base_cache): every run's base is evaluated in full every round (A-B check: same p / v bit for bit)
    bool sib_cache_valid = false;    // false: the trees changed outside the search rounds (reset, advance, refill): tags are cleared first
    float* part_w = nullptr;         // fp32 partials of the K-split window tiles: [7][part_w_rows][512]
    size_t part_w_rows = 0;
    int n_cu = 256;                  // compute units of the device
    int n_cu_all = 256;              // the same, set for every board size (n_cu above only with the sibling buffers)
    int mx_sw = 0;            // fc0 weights: fp8 copies are w * 2^mx_sw (hi) and (w - f16(w)) * 2^(mx_sw + 11) (lo)
    // fc0 operand format: the correction terms hi*lo + lo*hi of fc0 run either on block-scaled fp6 operands (FC0_FP6: 4
    // significant bits, products good to ~2^-15) or on f16 operands (FC0_F16: three f16 MFMAs per product, ~2^-22).  net_commit packs the
    // weights for both and, with fc0_policy = FC0_AUTO, measures both against the fp32 kernels on a fixed probe set and keeps the
    // faster one (fp6) only if its worst |dp|, |dv| stay within NET_PROBE_LIMIT (3e-4 against the 1e-3 contract).
    int fc0_policy = FC0_AUTO; // FC0_AUTO / FC0_FP6 / FC0_F16 (forced)
    int fc0_fmt = 0;          // format of FULL operand rows in use: FC0_FP6 or FC0_F16
    bool diff_fp6 = false;    // FC0_MIXED: fc0_fmt == FC0_F16, but the difference rows of sibling rounds (and their window tiles) are in the fp6 format
    void* wt_fc0x = nullptr;  // fc0 weights for FC0_F16: [half-step][stage g][m-tile i][hi s0, hi s1, lo s0, lo s1][lane][8] f16
    size_t row_u4_fmt[2] = {0, 0}; // a_fc0 row stride per format (row_u4 = the one in use)
    float probe[24] = {};     // commit-time probe: [0] plain rows, [1] |dp| fp6, [2] |dv| fp6, [3] |dp| f16, [4] |dv| f16, [5] max |logit| (fp32), [6] 1 = measured,
                              // [7] |dlogit| fp6, [8] |dlogit| f16 (plain rows); [9] rows of the synthetic sibling round checked, [10..18] its |dp|, |dv|, |dlogit| in fp6 / mixed / f16
    size_t bytes = 0;         // device bytes held
};

// sizes
inline int64_t net_tensor_size(int n, int idx) {
    const int64_t hw = (int64_t)n * n;
    if (idx == 0) return 3 * NC;
    if (idx == 1) return NC;
    if (idx >= 2 && idx < 23) {
        const int64_t q[7] = {NC * NM, NM, 9 * NM, NM * NM, NM, NM * NC, NC};
        return q[(idx - 2) % 7];
    }
    switch (idx) {
        case 23: return NC * hw * NF;
        case 24: return NF;
        case 25: return (int64_t)NF * NF;
        case 26: return NF;
        case 27: return NF;
        case 28: return 1;
        case 29: return NF * hw;
        case 30: return hw;
    }
    return -1;
}

// net_kernels.hip
// Forward of the `count` samples described by S.req_ref/req_aux (count read from S.d_count on the
// device; grids are sized for max_count).  Results in net.p / net.v.
// sibling_side >= 0: the rows are the requests of a search round of that side's trees (S.ts / S.req_node describe them): runs of
// sibling requests take the incremental trunk path.  -1: plain rows (mirror evaluations, shared-tree rounds).
// true if a forward of max_count rows leaves ALL its logits in net_logits() (split-precision modes, not chunked): skip_softmax may be used
bool net_logits_cover_batch(const Net& net, int max_count);
// true if net_forward_requests(net, S, max_count, ..., sibling_side >= 0) will group the requests by parent (k_group): the caller may then hand
// the zeroing of net.d_gcnt and the request-list fill to it (Net::gcnt_zeroed, Net::fill_in_group)
bool net_round_takes_sibling_path(const Net& net, int max_count);
constexpr int NET_GCNT_P0 = 8 + 81 + 7 + 16;   // Net::d_gcnt[NET_GCNT_P0 + pixel]: children whose stone lands in net pixel `pixel` (their rows take consecutive slots inside the window bin)
constexpr int NET_GCNT_INTS = NET_GCNT_P0 + 225; // ints of Net::d_gcnt
#ifndef SIDE_STREAM_DEFAULT
#define SIDE_STREAM_DEFAULT 0 // 1: Net::side_on unless OMOK_SIDE_STREAM=0; 0: only with OMOK_SIDE_STREAM=1.  Measured (profiles/r06_ab_side_stream.txt): three plies of configs[1] 0.3152 s off / 0.3145 s on --
                              // the full rows' fc0 (82 us per round) does start in the children kernel's tail, but the tail is one pass long and the join costs what it saves: off
#endif
// Net::d_work: [DIFF + 0..2] runs, single rows, rows in runs of the rounds on the difference path, [COPY + 0..2] the same on the copy path, [DIFF_FULL] runs of the difference
// path whose base was evaluated in full (base-cache misses + uncacheable runs), [WIN_PIXELS] window pixels the fc0 window tiles walked (tiles x their rectangles),
// [WIN_TILES] window tiles, [FULL_TILES] 128-row tiles of the full-row fc0
constexpr int NET_WORK_DIFF = 0, NET_WORK_COPY = 3, NET_WORK_DIFF_FULL = 6, NET_WORK_WIN_PIXELS = 7, NET_WORK_WIN_TILES = 8, NET_WORK_FULL_TILES = 9, NET_WORK_COUNT = 16;
// skip_softmax (split-precision modes only): stop behind the heads; the caller turns net_logits() into p / v itself (launch_softmax_scatter).
void net_forward_requests(Net& net, const Store& S, int max_count, hipStream_t st, struct Prof* prof, int sibling_side = -1, bool skip_softmax = false);
// Forward of explicit f32 inputs already in net.in_f32 ([count][3HW]); count is a host value
// and must also be stored in S.d_count[0] by the caller.
void net_forward_inputs(Net& net, const Store& S, int count, hipStream_t st, struct Prof* prof);
// The trees were changed outside the search rounds (reset, advance, refill, externally placed moves): cached base evaluations are void.
inline void net_invalidate_sibling_cache(Net& net) { net.sib_cache_valid = false; }
// Packs the raw tensors into the MFMA operand layouts (host-side repack + upload), then (split-precision modes) chooses fc0's operand
// format: see Net::fc0_policy.  S: the engine's store (the probe forwards use S.d_count).
int net_commit(Net& net, const Store& S, hipStream_t st);
void net_set_fc0_format(Net& net, int fmt); // FC0_FP6 / FC0_F16: switches row strides and kernels (cached base evaluations are void)
// pre-softmax policy logits of the LAST forward (rows of the last chunk in OMOK_NET_F32 mode): pointer and row stride in floats
const float* net_logits(const Net& net, int* row_stride);
size_t net_alloc(Net& net); // allocates buffers for net.max_b; returns bytes, 0 on failure
void net_free(Net& net);

// ---- tiny profiler: HIP-event pairs per category, resolved lazily ----
enum { PC_ROUND = 0, PC_TREE_OTHER, PC_TRUNK, PC_FC0, PC_TAIL, PC_PLY, PC_COUNT };
constexpr int PC_COUNT_MAX = 8;
// An event record between two kernels costs ~5 us of idle queue (measured: 10.6 us per category boundary = end + begin); with `every` = N > 1
// only every Nth search round is timed and the stats scale the sampled sums by rounds seen / rounds timed (ply-level work is always timed).
struct Prof {
    bool enabled = false;
    int every = 1;              // time 1 search round in `every`
    long long rounds_seen = 0, rounds_timed = 0;
    bool active = true;         // events are recorded now
    bool in_round = false;      // ... inside a (sampled) search round: sums go to the scaled accumulators
    double ms_s[PC_COUNT_MAX] = {};
    long long launches_s[PC_COUNT_MAX] = {};
    void round_begin() { in_round = true; active = every <= 1 || rounds_seen % every == 0; ++rounds_seen; if (active) ++rounds_timed; }
    void round_end() { in_round = false; active = true; }
    double scale() const { return rounds_timed > 0 ? (double)rounds_seen / (double)rounds_timed : 1.0; }
    double total_ms(int cat) const { return ms[cat] + ms_s[cat] * scale(); }
    double total_launches(int cat) const { return (double)launches[cat] + (double)launches_s[cat] * scale(); }
    struct Item { int cat; bool sampled; hipEvent_t a, b; };
    Item* items = nullptr;
    int n_items = 0, cap_items = 0;
    hipEvent_t* pool = nullptr;
    int n_pool = 0, cap_pool = 0;
    double ms[PC_COUNT] = {};
    long long launches[PC_COUNT] = {};
    hipEvent_t get();
    void begin(int cat, hipStream_t st);
    void end(hipStream_t st);
    void resolve(); // synchronises on the recorded events and accumulates ms
    void destroy();
};

} // namespace omok
