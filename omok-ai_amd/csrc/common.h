// common.h — data layout in HBM and launch interface shared by the HIP translation units.
//
// One tree = one arena, owned by exactly one wavefront at a time.  Layout:
//   tree id t = side * G + game            (side 0 = black agent, 1 = white agent, trainer.rs:83-84)
//   (stride_nodes / stride_tables = the capacities rounded up to odd numbers: Store)
//   hdr    [T][stride_nodes]        16 B   NodeHdr  (Node fields of mcts/src/node.rs:10-21)
//   board  [T][stride_nodes][2*NW]     u64    black words, white words (bit a = cell a)
//   policy [T][stride_nodes][ROWP]     f32    BoardState.policy (alpha-zero/src/mcts_node.rs:10)
//   child tables [T][stride_tables][ROWP]: cn u32, cw f32, cidx u16, corder u8; owner [T][stride_tables]
//     a node's (n, w) live in its PARENT's table at index = action (coalesced PUCT scan);
//     child.p is parent.policy[action] (the reference keeps them equal at all times).
//   ROWP = HW rounded up to 64 (one wave iteration per 64 cells); pad cells: corder = 0xFF.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace omok {

constexpr uint16_t NONE16 = 0xFFFFu;
constexpr uint8_t NONE8 = 0xFFu;
constexpr int KMAX = 64;
constexpr float F32_EPS = 1.1920928955078125e-7f;

enum { ST_IN_PROGRESS = 0, ST_DRAW = 1, ST_BLACK_WIN = 2, ST_WHITE_WIN = 3 };
enum { RNG_EXPAND = 1, RNG_NOISE = 2, RNG_SAMPLE = 3 };

template <int N>
struct Geo {
    static constexpr int HW = N * N;
    static constexpr int ROWP = (HW + 63) / 64 * 64;
    static constexpr int NW = (HW + 63) / 64;
    static constexpr int IT = ROWP / 64;
};

struct __attribute__((aligned(16))) NodeHdr {
    uint16_t parent;
    uint16_t table;
    uint16_t legal;
    uint16_t nch;
    uint8_t action;
    uint8_t status;
    uint8_t turn;
    uint8_t has_policy;
    uint32_t pad;
};
static_assert(sizeof(NodeHdr) == 16, "NodeHdr must be 16 bytes");

struct __attribute__((aligned(16))) TreeState {
    uint32_t n_nodes;
    uint32_t n_tables;
    uint32_t root_n;
    float root_w;
    uint32_t error;
    uint32_t n_req;
    uint32_t req_base;
    uint32_t pad;
};

struct GameState {
    uint8_t alive;
    uint8_t status;
    uint8_t external; // the pending move (last_action) was supplied by the caller (omok_play_actions), not sampled
    uint8_t harvested; // slots mode: the finished game's records have been packed out, the slot may take a new game
    int32_t plies;       // moves played
    int32_t last_action;
    int32_t mirror_idx;
    int32_t rp_len;      // transitions recorded = moves SAMPLED so far (turn_counts[index], src/trainer.rs:85,139,148)
    int32_t gid;         // game index of the slot's game (+ cfg.game_offset = the global id that keys its RNG streams); = the slot after a reset
    int32_t pad2[2];
};
static_assert(sizeof(GameState) == 32, "GameState must be 32 bytes");

struct Store {
    NodeHdr* hdr;
    uint64_t* board;
    float* policy;
    uint32_t* tcn;
    float* tcw;
    uint16_t* tcidx;
    uint8_t* tcorder;
    uint16_t* towner;
    TreeState* ts;
    uint16_t* req_node; // [T][KMAX]
    GameState* gs;      // [G]
    // dense request list of the current NN batch
    uint32_t* req_ref;  // (t << 16) | node
    uint32_t* req_aux;  // 0xFFFFFFFF none, else action | mode << 16 : stone to add before encoding
    int32_t* d_count;   // [0] = live batch size
    // replay
    uint64_t* rp_board; // [G][HW][2*NW]
    uint8_t* rp_turn;   // [G][HW]
    float* rp_pi;       // [G][HW][ROWP]
    float* rp_z;        // [G][HW]
    unsigned long long* d_bytes; // kernel-counted algorithmic bytes
    int cap_nodes, cap_tables, games;
    // Arena STRIDES per tree, in nodes / tables (>= the capacities): odd, so that the same node index of consecutive trees does not fall on the same HBM
    // channels.  With stride = capacity = 4224 (a multiple of 128: policy rows of 1 KiB -> a tree stride of 33 x 128 KiB) the policy scatter of a round -- 4096
    // trees writing the rows of about the same node indices -- ran at half the speed it has with 4225 (profiles/r04_arena_stride.txt).
    int stride_nodes, stride_tables;
};

struct RoundArgs {
    int side, round, K, ply;
    float epsilon, alpha;
    uint64_t seed; // Philox key of the current episode: cfg.seed + episode * 0x9E3779B97F4A7C15 (DESIGN.md "RNG contract")
    int64_t game_offset;
    const float* scatter_v; // non-NULL: the PREVIOUS round's backups (k_scatter with these values) run at the head of this round's kernel
};

// ---- tree_kernels.hip launchers (all asynchronous on `st`) ---------------------------------
void launch_reset(int n, const Store& S, const float* root_policy_dev /*ROWP*/, hipStream_t st);
// slots mode (omok_selfplay_run_slots): finished games are packed out (records appended at *out_count, per-game meta by game index) and their
// slots restarted with the next game indices while any are left
struct SlotMeta { long long offset; int32_t len; int32_t status; }; // per game index: first record, records, final GameStatus
void launch_harvest(int n, const Store& S, uint8_t* mask_dev /*[G]*/, long long* slot_off_dev /*[G]*/, long long* out_count_dev, SlotMeta* meta_dev,
                    uint8_t* dst_dev, long long cap_records, hipStream_t st);
void launch_refill(int n, const Store& S, const float* root_policy_dev, int32_t* next_gid_dev, int total_games, int32_t* new_gid_dev /*[G]*/, hipStream_t st);
void launch_round(int n, const Store& S, const RoundArgs& a, hipStream_t st);
// evals_dev[0] += the round's requests; zero_ptr[0 .. zero_n) = 0 (counters the round's forward expects zeroed); fill = false: the dense
// (tree, node) list is written by the net's grouping kernel instead (run-loop rounds on the sibling path: k_group visits every request anyway)
void launch_fill(const Store& S, int side, int K, hipStream_t st); // the dense request list alone (k_fill): what launch_scan(fill = true) appends
void launch_scan(int n, const Store& S, int side, int K, hipStream_t st, unsigned long long* evals_dev = nullptr, int32_t* zero_ptr = nullptr, int zero_n = 0,
                 bool fill = true);
// one tree searched by `waves` waves (MCTSExecutor::run): sh_req [waves][KMAX] u16, sh_cnt [2 * KMAX] u32 (counts | bases)
// rec_order / rec_pos non-NULL: RECORDED mode -- every simulation (round kernel) / every backup (scatter kernel) runs under the tree lock and
// appends its wave's index to rec_order[(*rec_pos)++]: an exactly replayable interleaving (omok_execute_shared_recorded)
void launch_round_shared(int n, const Store& S, const RoundArgs& a, int rounds_total, int group, int waves, uint16_t* sh_req, uint32_t* sh_cnt,
                         hipStream_t st, uint8_t* rec_order = nullptr, uint32_t* rec_pos = nullptr);
void launch_scatter_shared(int n, const Store& S, int side, const float* p_dev, const float* v_dev, int max_count, int waves, const uint16_t* sh_req,
                           const uint32_t* sh_cnt, hipStream_t st, uint8_t* rec_order = nullptr, uint32_t* rec_pos = nullptr);
constexpr int MAX_TREE_WAVES = 16; // one workgroup of 1024 threads
constexpr int MAX_GAMES = 32767;         // games per engine (omok_create)
constexpr int SCAN_GAMES_PER_THREAD = 32; // k_scan: one workgroup of 1024 threads, each owning this many consecutive games
// backups = false: the policies only; the backups (k_scatter) are deferred to launch_backups or into the next round's kernel (RoundArgs::scatter_v)
void launch_scatter(int n, const Store& S, int side, const float* p_dev, const float* v_dev, int max_count, hipStream_t st, bool backups = true);
void launch_backups(int n, const Store& S, int side, const float* v_dev, hipStream_t st);
// the same from the net's logits (softmax / tanh of the split-precision path fused into the policy scatter): writes v, vpre, the trees
void launch_softmax_scatter(int n, const Store& S, int side, const float* logits_dev, int lrow, float* v_dev, float* vpre_dev, int max_count,
                            hipStream_t st, bool backups = true);
void launch_sample(int n, const Store& S, int side, int ply, float temperature, int threshold, uint64_t seed,
                   int64_t game_offset, int32_t* actions_dev, hipStream_t st);
void launch_mirror_scan(int n, const Store& S, int side, hipStream_t st);
void launch_advance(int n, const Store& S, int side, const float* p_dev, hipStream_t st);
void launch_encode_requests(int n, const Store& S, float* out_dev /*[B][3HW]*/, int max_b, hipStream_t st);
void launch_env_play(int n, const int32_t* moves_dev, int batch, int len, int32_t* status_dev, uint8_t* boards_dev,
                     uint8_t* turns_dev, uint16_t* legal_dev, hipStream_t st);
void launch_encode_boards(int n, const uint8_t* boards_dev, const uint8_t* turns_dev, int batch, int mode,
                          float* out_dev, hipStream_t st);
void launch_replay_pack(int n, const Store& S, const long long* offsets_dev, uint8_t* dst_dev, long long cap_records, hipStream_t st);
void launch_replay_offsets(int n, const Store& S, int per_transition, long long* offsets_dev /*[games + 1]*/, hipStream_t st);
void launch_set_actions(int n, const Store& S, int side, const int32_t* actions_dev, uint32_t* d_flags /*[0] illegal, [1] missing*/, hipStream_t st);
void launch_clear_actions(const Store& S, hipStream_t st);
void launch_policy(int n, const Store& S, int side, float* pi_dev /*[G][HW]*/, uint8_t* has_dev /*[G]*/, hipStream_t st);
void launch_env_place(int n, uint8_t* boards_dev, uint8_t* turns_dev, uint16_t* legal_dev, const int32_t* actions_dev, int batch,
                      int32_t* status_dev, hipStream_t st);
int set_advance_lds_attribute(int n, size_t bytes);
void launch_replay_augment(int n, const Store& S, const long long* offsets_dev, int game_first, int game_count, long long base_sub,
                           uint8_t* dst_dev, long long cap_records, hipStream_t st);
size_t advance_lds_bytes(int cap_nodes, int cap_tables);

// ---- net_kernels.hip --------------------------------------------------------------------------
struct NetBuffers; // defined in net.h
} // namespace omok
