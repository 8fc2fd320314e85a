// tree_kernels.hip — board rules, PUCT select / expand / backup, scatter, action sampling and
// re-rooting as hand-written gfx950 kernels.  ONE WAVEFRONT (64 lanes) OWNS ONE TREE.
//
// What each kernel replaces in the reference (AcrylicShrimp/omok-ai):
//   k_round    alpha-zero/src/parallel_mcts_executor.rs:44-192 (generate_requests: root Dirichlet
//              noise :48-76, K x { select_leaf mcts/src/node.rs:39-59 with PUCT pme.rs:81-90,
//              277-286; terminal shortcut :92-97; random untried action :101-125; place_stone
//              environment/src/lib.rs:104-166; expand node.rs:61-81; propagate node.rs:83-99 })
//   k_scan     the order-preserving request concat of pme.rs:194-205 (tree order, then sim order)
//   k_scatter  pme.rs:222-265
//   k_sample   alpha-zero/src/agent.rs:43-137 + src/trainer.rs:138-173 (transition record)
//   k_advance  agent.rs:144-232 + mcts/src/lib.rs:47-93 (transition) + trainer.rs:156-201
//
// Compiled with -ffp-contract=off: every f32 op below must round exactly like the Rust
// expression it restates (no FMA contraction), div/sqrt are the correctly rounded forms.
// Work split inside a wave: lanes sweep rows (cell a = 64*j + lane); scalar bookkeeping is
// computed uniformly by all lanes and stored by lane 0.  A __syncthreads() (single-wave
// workgroup: a fence + waitcnt) separates global stores from later cross-lane loads.
#include "common.h"
#include <type_traits>

namespace omok {

#define LANE ((int)(threadIdx.x & 63))

// ---------------------------------------------------------------------------------------------
// RNG contract (Philox4x32-10 + fixed-order f64 log/exp): must match DESIGN.md "RNG contract".
// ---------------------------------------------------------------------------------------------
struct U4 { uint32_t x, y, z, w; };

__device__ inline U4 philox(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3) {
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t h0 = __umulhi(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
        const uint32_t h1 = __umulhi(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
        c0 = n0; c1 = l1; c2 = n2; c3 = l0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return U4{c0, c1, c2, c3};
}

__device__ inline double det_log(double x) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(x);
    int e = (int)((b >> 52) & 0x7ff) - 1023;
    double m = __longlong_as_double((long long)((b & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL));
    if (m > 1.4142135623730951) { m = m * 0.5; e = e + 1; }
    const double s = (m - 1.0) / (m + 1.0);
    const double s2 = s * s;
    double poly = 1.0 / 23.0;
    poly = poly * s2 + 1.0 / 21.0;
    poly = poly * s2 + 1.0 / 19.0;
    poly = poly * s2 + 1.0 / 17.0;
    poly = poly * s2 + 1.0 / 15.0;
    poly = poly * s2 + 1.0 / 13.0;
    poly = poly * s2 + 1.0 / 11.0;
    poly = poly * s2 + 1.0 / 9.0;
    poly = poly * s2 + 1.0 / 7.0;
    poly = poly * s2 + 1.0 / 5.0;
    poly = poly * s2 + 1.0 / 3.0;
    poly = poly * s2 + 1.0;
    const double lm = 2.0 * s * poly;
    return (double)e * 0.6931471805599453 + lm;
}

__device__ inline double pow2i(int k) { return __longlong_as_double((long long)(1023 + k) << 52); }

__device__ inline double det_exp(double x) {
    if (x > 709.0) return __longlong_as_double(0x7ff0000000000000LL);
    if (x < -745.0) return 0.0;
    const double t = x * 1.4426950408889634 + 0.5;
    long long ki = (long long)t;
    if ((double)ki > t) ki = ki - 1;
    const double k = (double)ki;
    const double r = (x - k * 0.6931471803691238) - k * 1.9082149292705877e-10;
    double p = 1.0 / 6227020800.0;
    p = p * r + 1.0 / 479001600.0;
    p = p * r + 1.0 / 39916800.0;
    p = p * r + 1.0 / 3628800.0;
    p = p * r + 1.0 / 362880.0;
    p = p * r + 1.0 / 40320.0;
    p = p * r + 1.0 / 5040.0;
    p = p * r + 1.0 / 720.0;
    p = p * r + 1.0 / 120.0;
    p = p * r + 1.0 / 24.0;
    p = p * r + 1.0 / 6.0;
    p = p * r + 0.5;
    p = p * r + 1.0;
    p = p * r + 1.0;
    int kk = (int)ki;
    if (kk < -1000) {
        p = p * pow2i(-1000);
        kk = kk + 1000;
    }
    return p * pow2i(kk);
}

__device__ inline float det_expf(float x) { return (float)det_exp((double)x); }
__device__ inline double u01(uint32_t x) { return ((double)x + 0.5) * 2.3283064365386963e-10; }

// Gamma(alpha,1) draw of one cell: Ahrens-Dieter GS for the fractional part + unit exponentials.
__device__ float gamma_draw(float alpha_f, uint64_t seed, uint32_t cell, uint32_t ply, uint32_t tree_global) {
    const double alpha = (double)alpha_f;
    int k = (int)alpha;
    if (k > 32) k = 32;
    const double f = alpha - (double)k;
    double x = 0.0;
    if (f > 0.0) {
        const double b = 1.0 + f * 0.36787944117144233;
        double g = 0.0;
        for (uint32_t attempt = 0; attempt < 200; ++attempt) {
            const U4 o = philox(seed, cell * 256u + attempt, ply, tree_global, RNG_NOISE);
            const double u1 = u01(o.x), u2 = u01(o.y);
            const double p = b * u1;
            if (p <= 1.0) {
                const double c = det_exp(det_log(p) / f);
                if (u2 <= det_exp(-c)) { g = c; break; }
            } else {
                const double c = -det_log((b - p) / f);
                if (u2 <= det_exp((f - 1.0) * det_log(c))) { g = c; break; }
            }
        }
        x = g;
    }
    for (int i = 0; i < k; ++i) {
        const U4 o = philox(seed, cell * 256u + 255u - (uint32_t)i, ply, tree_global, RNG_NOISE);
        x = x + (-det_log(u01(o.x)));
    }
    if (x < 1e-30) x = 0.0;
    return (float)x;
}

// ---------------------------------------------------------------------------------------------
// wave helpers
// ---------------------------------------------------------------------------------------------
#ifndef WAVE_MAX_DPP
#define WAVE_MAX_DPP 1 // (A-B builds: 0 = the butterfly of 64-bit shuffles, 12 ds_bpermute in a dependent chain; same result)
#endif
// unsigned maximum over the 64 lanes by data-parallel-primitive moves (no LDS crossbar): inclusive max-scan inside each row of 16 lanes (row_shr 1, 2, 4, 8; lanes without
// a source keep the identity 0), lane 15 of rows 0 / 2 into rows 1 / 3 (row_bcast15), lane 31 into rows 2 and 3 (row_bcast31): lane 63 holds the wave's maximum.
__device__ inline uint32_t wave_max_u32_dpp(uint32_t x) {
    auto step = [](uint32_t v, auto ctrl, auto row_mask) {
        const uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, decltype(ctrl)::value, decltype(row_mask)::value, 0xf, false);
        return o > v ? o : v;
    };
    x = step(x, std::integral_constant<int, 0x111>{}, std::integral_constant<int, 0xf>{});
    x = step(x, std::integral_constant<int, 0x112>{}, std::integral_constant<int, 0xf>{});
    x = step(x, std::integral_constant<int, 0x114>{}, std::integral_constant<int, 0xf>{});
    x = step(x, std::integral_constant<int, 0x118>{}, std::integral_constant<int, 0xf>{});
    x = step(x, std::integral_constant<int, 0x142>{}, std::integral_constant<int, 0xa>{});
    x = step(x, std::integral_constant<int, 0x143>{}, std::integral_constant<int, 0xc>{});
    return (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
}
__device__ inline unsigned long long wave_max_u64(unsigned long long v) {
#if WAVE_MAX_DPP
    // lexicographic: the maximum of the high words, then the maximum of the low words among the lanes that hold it (a lane with v = 0 never wins against a candidate)
    const uint32_t hi = (uint32_t)(v >> 32), lo = (uint32_t)v;
    const uint32_t H = wave_max_u32_dpp(hi);
    const uint32_t L = wave_max_u32_dpp(hi == H ? lo : 0u);
    return ((unsigned long long)H << 32) | (unsigned long long)L;
#else
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long x = __shfl_xor(v, o, 64);
        v = x > v ? x : v;
    }
    return v;
#endif
}
__device__ inline uint32_t wave_sum_u32(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ inline uint32_t total_key_biased(float f) { // f32::total_cmp order as an unsigned key
    int b = __float_as_int(f);
    b ^= (int)(((unsigned)(b >> 31)) >> 1);
    return (uint32_t)b ^ 0x80000000u;
}
__device__ inline int nth_set_bit(unsigned long long m, int r) { // index of the r-th (0-based) set bit of m; m, r wave-uniform, all 64 lanes call
    // lane l: is bit l set and preceded by exactly r set bits?  (a loop clearing the lowest bit r times was ~100 scalar instructions per
    //  simulation in k_round, whose cached simulations are bound by instruction issue)
    const int l = (int)LANE;
    const bool hit = ((m >> l) & 1ULL) && __popcll(m & ((1ULL << l) - 1ULL)) == r;
    return __ffsll((long long)__ballot(hit)) - 1;
}
// (Both helpers touch EVERY word with a computed mask: a chain of selects on the word index is turned back into an indexed
//  access by the compiler, which then keeps the bitboards in scratch memory -- 120 B per lane, ~6 KB of HBM writes per
//  simulation in k_round, the rocprofv3 WRITE_SIZE of round 1.)
template <int NW>
__device__ inline int get_bit(const uint64_t* w, int c) {
    const int wi = c >> 6, sh = c & 63;
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < NW; ++i) r |= (uint32_t)((w[i] >> sh) & 1ULL) & (wi == i ? 1u : 0u);
    return (int)r;
}
template <int NW>
__device__ inline void set_bit(uint64_t* w, int c) {
    const int wi = c >> 6;
    const uint64_t bit = 1ULL << (c & 63);
#pragma unroll
    for (int i = 0; i < NW; ++i) w[i] |= wi == i ? bit : 0ULL;
}

// sequential f32 sum of s[0..n) in ascending order (Rust iter().sum::<f32>()), by every lane
// redundantly from LDS so the result is wave-uniform without a broadcast
__device__ inline float seq_sum(const float* s, int n) {
    float acc = 0.0f;
    for (int i = 0; i < n; ++i) acc += s[i];
    return acc;
}

// The same sequential sum over a row held in REGISTERS (x[j] of lane l = element 64 j + l): element by element through v_readlane,
// ascending -- the same additions in the same order as seq_sum, without 225 dependent LDS round trips (~16 k cycles per row, the
// whole run time of k_scatter_policy).  Wave-uniform result.
template <int N>
__device__ inline float seq_sum_regs(const float (&x)[Geo<N>::IT]) {
    float acc = 0.0f;
#pragma unroll
    for (int j = 0; j < Geo<N>::IT; ++j)
#pragma unroll
        for (int l = 0; l < 64; ++l)
            if (j * 64 + l < Geo<N>::HW) acc += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x[j]), l));
    return acc;
}

// ---------------------------------------------------------------------------------------------
// per-tree view
// ---------------------------------------------------------------------------------------------
#ifdef KROUND_PROF // diagnostic build: per launch and tree, shader-clock cycles per phase of k_round
__device__ unsigned long long g_kprof[64][8192][8];
__device__ unsigned int g_klaunch;
#define KP(i) do { const unsigned long long now_ = __builtin_readcyclecounter(); kp[i] += now_ - kp_last; kp_last = now_; } while (0)
#else
#define KP(i) do { } while (0)
#endif
#ifndef PARALLEL_PICKS
#define PARALLEL_PICKS 1 // (A-B builds: 0 = run_sims picks its cells and checks its children one at a time, as in round 4; same results)
#endif
template <int N>
struct Tree {
    using G = Geo<N>;
    NodeHdr* hdr;
    uint64_t* board;
    float* pol;
    uint32_t* cn;
    float* cw;
    uint16_t* cidx;
    uint8_t* corder;
    uint16_t* owner;
    TreeState* ts;
    uint16_t* req;
    __device__ Tree(const Store& S, int t) {
        const size_t tn = (size_t)t * (size_t)S.stride_nodes, tt = (size_t)t * (size_t)S.stride_tables;
        hdr = S.hdr + tn;
        board = S.board + tn * (2 * G::NW);
        pol = S.policy + tn * G::ROWP;
        cn = S.tcn + tt * G::ROWP;
        cw = S.tcw + tt * G::ROWP;
        cidx = S.tcidx + tt * G::ROWP;
        corder = S.tcorder + tt * G::ROWP;
        owner = S.towner + tt;
        ts = S.ts + t;
        req = S.req_node + (size_t)t * KMAX;
    }
};

struct Regs { // wave-uniform running state of a tree
    uint32_t n_nodes, n_tables, root_n, n_req, error;
    float root_w;
    unsigned long long bytes;
    // select_leaf memo: between two backups nothing that the PUCT descent reads changes (an expansion only adds a child to
    // the leaf, whose children are not scanned while it has untried actions), so the next simulation of the round reaches
    // the same leaf.  memo_node = that leaf (-1 = none), memo_n = its n as stored in its parent's table, memo_bytes = the
    // algorithmic bytes of the descent it stands for (the counter keeps the reference algorithm's meaning).
    int memo_node = -1;
    uint32_t memo_n = 0;
    unsigned long long memo_bytes = 0;
    bool dirty = false; // this wave has stores in flight that a later load of its own may depend on (see LeafCache)
#ifdef KROUND_PROF
    unsigned long long kp[8] = {}, kp_last = 0;
#endif
};

// The memoised leaf in registers.  Consecutive simulations of a round expand the SAME leaf (see memo_node): after the first one the wave
// knows the leaf's header, board and which of its cells are still untried -- it wrote the changes itself -- so the following
// simulations need no load at all, and without a load of its own stores the wave needs no store -> load fence per simulation either
// (each was a round trip to L2).  Any path that does read the tree (descent, backup, a new leaf) first waits for the stores in flight
// (Regs::dirty).  The arithmetic and its order are unchanged.
template <int N>
struct LeafCache {
    int node = -1;
    NodeHdr h;
    uint64_t bb[2 * Geo<N>::NW];
    unsigned long long cand[Geo<N>::IT];
};
__device__ inline void fence_own_stores(Regs& R) {
    if (R.dirty) { __syncthreads(); R.dirty = false; }
}

// environment/src/lib.rs:104-166 on bitboards, wave-cooperative (all 64 lanes must call).
// `own` already contains the new stone.  Returns 1 if any of the 4 lines is EXACTLY five.
template <int N>
__device__ inline int exactly_five(const uint64_t* own, int a) {
    constexpr int NW = Geo<N>::NW;
    const int lane = LANE;
    const int d = lane / 5, k = lane % 5 + 1;
    // ray order: (-1,0) (1,0) | (0,-1) (0,1) | (-1,-1) (1,1) | (-1,1) (1,-1)   (lib.rs:112-145)
    const int dx = (d == 0 || d == 4 || d == 6) ? -1 : ((d == 1 || d == 5 || d == 7) ? 1 : 0);
    const int dy = (d == 2 || d == 4 || d == 7) ? -1 : ((d == 3 || d == 5 || d == 6) ? 1 : 0);
    bool hit = false;
    if (lane < 40) {
        const int x = a % N + dx * k, y = a / N + dy * k;
        if (x >= 0 && x < N && y >= 0 && y < N) hit = get_bit<NW>(own, y * N + x) != 0;
    }
    const unsigned long long m = __ballot(hit);
    int run[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const uint32_t f = (uint32_t)(m >> (5 * i)) & 31u;
        run[i] = __ffs((int)(~f & 63u)) - 1; // consecutive hits from k=1 (lib.rs:179-190)
    }
    return (1 + run[0] + run[1] == 5) | (1 + run[2] + run[3] == 5) | (1 + run[4] + run[5] == 5) |
           (1 + run[6] + run[7] == 5);
}


// clone + place_stone on bitboards (static register indexing only).  bb = black NW | white NW.
template <int N>
__device__ inline int place_and_status(uint64_t* bb, int turn, int legal_before, int action) {
    constexpr int NW = Geo<N>::NW;
    uint64_t mine[NW];
#pragma unroll
    for (int i = 0; i < NW; ++i) mine[i] = turn == 0 ? bb[i] : bb[NW + i];
    set_bit<NW>(mine, action);
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        bb[i] = turn == 0 ? mine[i] : bb[i];
        bb[NW + i] = turn == 0 ? bb[NW + i] : mine[i];
    }
    const int five = exactly_five<N>(mine, action);
    return five ? (turn == 0 ? ST_BLACK_WIN : ST_WHITE_WIN) : (legal_before == 1 ? ST_DRAW : ST_IN_PROGRESS);
}

// mcts/src/node.rs:83-99.  All lanes walk the path (uniform loads); lane 0 stores.
template <int N>
__device__ inline void backup(const Tree<N>& T, Regs& R, int x, float v) {
    constexpr int ROWP = Geo<N>::ROWP;
    const int lane = LANE;
    while (x != 0) {
        const NodeHdr h = T.hdr[x];
        const size_t slot = (size_t)T.hdr[h.parent].table * ROWP + h.action;
        const uint32_t n = T.cn[slot] + 1u;
        const float w = T.cw[slot] + v;
        if (lane == 0) { T.cn[slot] = n; T.cw[slot] = w; }
        v = -v;
        x = h.parent;
        R.bytes += 16;
    }
    R.root_n += 1u;
    R.root_w += v;
    R.bytes += 16;
}

// Adds a child under `parent` (hp = its header as currently stored).  Returns the node index,
// or -1 on arena overflow.  Board words `bb` (black NW, white NW) are the child's board.
template <int N>
__device__ inline int add_child(const Store& S, const Tree<N>& T, Regs& R, int parent, NodeHdr& hp, int action,
                                const uint64_t* bb, int status, int turn, int has_policy) { // (hp is updated like the stored header: table, nch)
    using G = Geo<N>;
    const int lane = LANE;
    if (R.n_nodes >= (uint32_t)S.cap_nodes) { R.error |= 1u; return -1; }
    int tab = hp.table;
    if (tab == NONE16) {
        if (R.n_tables >= (uint32_t)S.cap_tables) { R.error |= 1u; return -1; }
        tab = (int)R.n_tables++;
#pragma unroll
        for (int j = 0; j < G::IT; ++j) T.corder[(size_t)tab * G::ROWP + j * 64 + lane] = NONE8;
        if (lane == 0) { T.owner[tab] = (uint16_t)parent; T.hdr[parent].table = (uint16_t)tab; }
        __syncthreads(); // corder row init must land before the slot store below
    }
    const int idx = (int)R.n_nodes++;
    if (lane == 0) {
        const size_t slot = (size_t)tab * G::ROWP + action;
        T.corder[slot] = (uint8_t)hp.nch;
        T.cidx[slot] = (uint16_t)idx;
        T.cn[slot] = 0u;
        T.cw[slot] = 0.0f;
        T.hdr[parent].nch = (uint16_t)(hp.nch + 1);
        NodeHdr c;
        c.parent = (uint16_t)parent;
        c.table = NONE16;
        c.legal = (uint16_t)(hp.legal - 1);
        c.nch = 0;
        c.action = (uint8_t)action;
        c.status = (uint8_t)status;
        c.turn = (uint8_t)turn;
        c.has_policy = (uint8_t)has_policy;
        c.pad = 0;
        T.hdr[idx] = c;
    }
    if (lane < 2 * G::NW) {
        uint64_t w = bb[0];
#pragma unroll
        for (int i = 1; i < 2 * G::NW; ++i) w = lane == i ? bb[i] : w;
        T.board[(size_t)idx * (2 * G::NW) + lane] = w;
    }
    R.bytes += 24 + 8 * 2 * G::NW;
    hp.table = (uint16_t)tab;
    hp.nch = (uint16_t)(hp.nch + 1);
    return idx;
}

// ---------------------------------------------------------------------------------------------
// k_round: Dirichlet noise (round 0) + K simulations for one tree
// ---------------------------------------------------------------------------------------------
template <int N>
__device__ void apply_noise(const Tree<N>& T, const RoundArgs& A, uint32_t tree_global, float* s_row) {
    using G = Geo<N>;
    const int lane = LANE;
    NodeHdr h0 = T.hdr[0];
    uint64_t bb[2 * G::NW];
#pragma unroll
    for (int i = 0; i < 2 * G::NW; ++i) bb[i] = T.board[i];
#pragma unroll
    for (int j = 0; j < G::IT; ++j) {
        const int a = j * 64 + lane;
        s_row[a] = a < G::HW ? gamma_draw(A.alpha, A.seed, (uint32_t)a, (uint32_t)A.ply, tree_global) : 0.0f;
    }
    __syncthreads();
    const float total = seq_sum(s_row, G::HW);
    __syncthreads();
    const bool ok = total > 0.0f;
    const float inv = ok ? __fdiv_rn(1.0f, total) : 0.0f;
    const float ph = h0.legal ? __fdiv_rn(1.0f, (float)h0.legal) : 0.0f;
#pragma unroll
    for (int j = 0; j < G::IT; ++j) {
        const int a = j * 64 + lane;
        float m = 0.0f;
        if (a < G::HW) {
            const float noise = ok ? s_row[a] * inv : __fdiv_rn(1.0f, (float)G::HW);
            const bool empty = !(((bb[j] | bb[G::NW + j]) >> lane) & 1ULL);
            const float p = h0.has_policy ? T.pol[a] : (empty ? ph : 0.0f);
            m = (1.0f - A.epsilon) * p + A.epsilon * noise; // pme.rs:59
        }
        s_row[a] = m;
    }
    __syncthreads();
    const float sum = seq_sum(s_row, G::HW);
    const float sum_inv = __fdiv_rn(1.0f, sum); // pme.rs:63-68 (no epsilon guard)
#pragma unroll
    for (int j = 0; j < G::IT; ++j) {
        const int a = j * 64 + lane;
        T.pol[a] = a < G::HW ? s_row[a] * sum_inv : 0.0f;
    }
    if (lane == 0 && !h0.has_policy) T.hdr[0].has_policy = 1;
    __syncthreads();
}

// What a descent FROM THE ROOT passed, level by level (wave-uniform, scalar registers): the table slot it took and that slot's (n, w) as the scan read them.  A simulation
// that ends on a terminal leaf is backed up along exactly these slots, and nothing writes them between the scan and the backup: the backup needs no walk and no load
// (k_round's slowest trees -- those that spend a round on root -> terminal descents -- spent a fifth of their time in that walk).
constexpr int PR_LEVELS = 3;
struct PathRec {
    int tab[PR_LEVELS], act[PR_LEVELS];
    uint32_t n[PR_LEVELS];
    float w[PR_LEVELS];
    int depth; // levels recorded by the last descent; < 0: it did not start at the root, > PR_LEVELS: deeper than the record
};
__device__ inline int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ inline NodeHdr uni(const NodeHdr& h) {
    uint4 q = __builtin_bit_cast(uint4, h);
    q.x = (uint32_t)uni((int)q.x); q.y = (uint32_t)uni((int)q.y); q.z = (uint32_t)uni((int)q.z); q.w = (uint32_t)uni((int)q.w);
    return __builtin_bit_cast(NodeHdr, q);
}

// select_leaf (node.rs:43-58) with the PUCT selector (pme.rs:81-90): from (node, h) down while the node is fully expanded; the last maximal child in
// insertion order wins (max_by).  Updates node, its header, its visit count as stored in its parent's table, and the algorithmic bytes of the path.
template <int N, bool REC>
__device__ inline void select_leaf_impl(const Tree<N>& T, NodeHdr& h, int& node, uint32_t& node_n, unsigned long long& path_bytes, PathRec& rec) {
    using G = Geo<N>;
    constexpr int ROWP = G::ROWP;
    const int lane = LANE;
    int lev = 0;
    while (h.nch == h.legal && h.nch != 0) {
        const uint32_t pn = node_n > 1u ? node_n : 1u;
        const float sq = __fsqrt_rn((float)pn);
        const size_t tb = (size_t)h.table * ROWP;
        const float ph = __fdiv_rn(1.0f, (float)h.legal); // placeholder prior of a not-yet-evaluated node
        unsigned long long best = 0ULL;
        // every load of the level goes out before the first use: rows are ROWP long (pad cells: rank NONE8; (n, w, p) of a cell without child are read and not used).
        // (Loaded under `if (ord != NONE8)` the scan was a chain of 2 x IT dependent round trips per level: k_round's slowest trees spent 3/4 of their time here.)
        uint8_t ordv[G::IT];
        uint32_t nv[G::IT];
        uint16_t civ[G::IT];
        float wv[G::IT], pv[G::IT];
#pragma unroll
        for (int j = 0; j < G::IT; ++j) {
            const int a = j * 64 + lane;
            ordv[j] = T.corder[tb + a];
            nv[j] = T.cn[tb + a];
            wv[j] = T.cw[tb + a];
            pv[j] = h.has_policy ? T.pol[(size_t)node * ROWP + a] : ph;
            civ[j] = T.cidx[tb + a];
        }
#pragma unroll
        for (int j = 0; j < G::IT; ++j) {
            const int a = j * 64 + lane;
            const uint8_t ord = ordv[j];
            if (ord != NONE8) {
                const uint32_t n = nv[j];
                const float w = wv[j];
                const float p = pv[j];
                const float q = __fdiv_rn(w, (float)n + F32_EPS);
                const float bias = __fdiv_rn(sq, (float)(1u + n));
                const float score = q + (1.0f * p) * bias;
                const unsigned long long v = ((unsigned long long)total_key_biased(score) << 32) |
                                             ((unsigned long long)ord << 16) | (unsigned long long)a;
                best = v > best ? v : best; // last max in insertion order wins ties (max_by)
            }
        }
        best = wave_max_u64(best);
        const int a_best = (int)(best & 0xFFFFu);
        path_bytes += 12ull * h.nch;
        { // the chosen child's n, w and index are in the registers of the lane that scanned it
            uint32_t n_sel = nv[0], c_sel = civ[0];
            float w_sel = wv[0];
#pragma unroll
            for (int j = 1; j < G::IT; ++j) {
                n_sel = (a_best >> 6) == j ? nv[j] : n_sel;
                c_sel = (a_best >> 6) == j ? (uint32_t)civ[j] : c_sel;
                w_sel = (a_best >> 6) == j ? wv[j] : w_sel;
            }
            node_n = (uint32_t)__shfl((int)n_sel, a_best & 63, 64);
            node = __shfl((int)c_sel, a_best & 63, 64);
            if (REC) {
                const float w_best = __shfl(w_sel, a_best & 63, 64);
#pragma unroll
                for (int l = 0; l < PR_LEVELS; ++l)
                    if (l == lev) {
                        rec.tab[l] = uni((int)h.table);
                        rec.act[l] = uni(a_best);
                        rec.n[l] = (uint32_t)uni((int)node_n);
                        rec.w[l] = __builtin_bit_cast(float, uni(__builtin_bit_cast(int, w_best)));
                    }
            }
        }
        h = T.hdr[node];
        lev += 1;
    }
    if (REC) rec.depth = lev;
}
template <int N>
__device__ inline void select_leaf(const Tree<N>& T, NodeHdr& h, int& node, uint32_t& node_n, unsigned long long& path_bytes) {
    PathRec none;
    select_leaf_impl<N, false>(T, h, node, node_n, path_bytes, none);
}

// Does a stone at cell `a` make exactly five with the `own` stones along direction pair `pr` (0: horizontal, 1: vertical, 2: (-1,-1)/(1,1), 3: (-1,1)/(1,-1))?
// The reference's run counts (environment/src/lib.rs:112-145, 179-190: consecutive own stones from the new one outwards, at most 5 per ray) for ONE pair of opposite rays,
// per lane: `a` and `pr` may differ between lanes, `own` is the mover's bitboard WITHOUT the new stone (the rays start next to it).  exactly_five() is the OR over pr.
template <int N>
__device__ inline bool five_in_pair(const uint64_t* own, int a, int pr) {
    constexpr int NW = Geo<N>::NW;
    const int dx = pr == 1 ? 0 : -1, dy = pr == 0 ? 0 : (pr == 3 ? 1 : -1);
    const int x0 = a % N, y0 = a / N;
    int run_a = 0, run_b = 0;
    bool on_a = true, on_b = true;
#pragma unroll
    for (int k = 1; k <= 5; ++k) {
        const int xa = x0 + dx * k, ya = y0 + dy * k, xb = x0 - dx * k, yb = y0 - dy * k;
        const bool in_a = xa >= 0 && xa < N && ya >= 0 && ya < N, in_b = xb >= 0 && xb < N && yb >= 0 && yb < N;
        on_a = on_a && in_a && get_bit<NW>(own, in_a ? ya * N + xa : 0) != 0;
        on_b = on_b && in_b && get_bit<NW>(own, in_b ? yb * N + xb : 0) != 0;
        run_a += on_a ? 1 : 0;
        run_b += on_b ? 1 : 0;
    }
    return 1 + run_a + run_b == 5;
}
// index of the r-th (0-based) set bit of w, r < popcount(w); per lane (w and r may differ between lanes)
__device__ inline int nth_set_bit_lane(uint64_t w, int r) {
    uint64_t cur = w;
    int base = 0;
#pragma unroll
    for (int sh = 32; sh >= 1; sh >>= 1) {
        const uint64_t low = cur & ((1ULL << sh) - 1ULL);
        const int c = __popcll(low);
        const bool up = r >= c;
        r -= up ? c : 0;
        base += up ? sh : 0;
        cur = up ? (cur >> sh) : low;
    }
    return base;
}

template <int N>
__device__ void run_sim(const Store& S, const Tree<N>& T, Regs& R, LeafCache<N>& C, const RoundArgs& A, uint32_t sim_index,
                        uint32_t tree_global) {
    using G = Geo<N>;
    constexpr int ROWP = G::ROWP, NW = G::NW;
    const int lane = LANE;
    int node = 0;
    uint32_t node_n = R.root_n;
    unsigned long long path_bytes = 0;
    if (R.memo_node >= 0) { // resume at the remembered leaf
        node = R.memo_node;
        node_n = R.memo_n;
        path_bytes = R.memo_bytes;
    } else C.node = -1;
    const bool cached = C.node == node;
    NodeHdr h;
    if (cached) h = C.h;
    else { fence_own_stores(R); h = T.hdr[node]; }
    if (h.nch == h.legal && h.nch != 0) { fence_own_stores(R); C.node = -1; } // the descent reads the tables this wave has been writing
    select_leaf<N>(T, h, node, node_n, path_bytes);
    R.bytes += path_bytes;
    R.memo_node = node;
    R.memo_n = node_n;
    R.memo_bytes = path_bytes;
    // ---- terminal leaf (pme.rs:92-97) ----
    if (h.status != ST_IN_PROGRESS) {
        fence_own_stores(R);
        backup<N>(T, R, node, h.status >= ST_BLACK_WIN ? 1.0f : 0.0f);
        R.memo_node = -1; // n / w changed along the path
        C.node = -1;
        __syncthreads();
        return;
    }
    // ---- random untried legal action (pme.rs:101-125) ----
    uint64_t bb[2 * NW];
    unsigned long long cand[G::IT];
    int total = 0;
    if (C.node == node) { // (still valid: the descent above did not move)
#pragma unroll
        for (int i = 0; i < 2 * NW; ++i) bb[i] = C.bb[i];
#pragma unroll
        for (int j = 0; j < G::IT; ++j) { cand[j] = C.cand[j]; total += __popcll(cand[j]); }
    } else {
        fence_own_stores(R);
#pragma unroll
        for (int i = 0; i < 2 * NW; ++i) bb[i] = T.board[(size_t)node * (2 * NW) + i];
#pragma unroll
        for (int j = 0; j < G::IT; ++j) {
            const int a = j * 64 + lane;
            bool c = a < G::HW && !(((bb[j] | bb[NW + j]) >> lane) & 1ULL);
            if (c && h.table != NONE16) c = T.corder[(size_t)h.table * ROWP + a] == NONE8;
            cand[j] = __ballot(c);
            total += __popcll(cand[j]);
        }
        C.node = node;
#pragma unroll
        for (int i = 0; i < 2 * NW; ++i) C.bb[i] = bb[i];
    }
    R.bytes += 2 * (G::HW / 8);
    if (total == 0) return; // "There's no action for now": the simulation is consumed
    const U4 o = philox(A.seed, sim_index, (uint32_t)A.ply, tree_global, RNG_EXPAND);
    int r = (int)__umulhi(o.x, (uint32_t)total);
    int action = 0;
    bool found = false;
#pragma unroll
    for (int j = 0; j < G::IT; ++j) {
        const int c = __popcll(cand[j]);
        if (!found) {
            if (r < c) { action = j * 64 + nth_set_bit(cand[j], r); found = true; }
            else r -= c;
        }
    }
    // ---- place the stone on a copy of the leaf's board (pme.rs:128-135) ----
    const int status = place_and_status<N>(bb, h.turn, h.legal, action);
    // ---- expand (node.rs:61-81); the placeholder policy of pme.rs:140-156 is implicit ----
    const int child = add_child<N>(S, T, R, node, h, action, bb, status, 1 - h.turn, 0);
    if (child < 0) { C.node = -1; return; }
    R.dirty = true;
    // the leaf as it is now: one more child, the cell no longer untried
    C.h = h;
#pragma unroll
    for (int j = 0; j < G::IT; ++j) C.cand[j] = cand[j] & ~((action >> 6) == j ? (1ULL << (action & 63)) : 0ULL);
    if (status != ST_IN_PROGRESS) { // pme.rs:177-181
        fence_own_stores(R);
        backup<N>(T, R, child, status == ST_DRAW ? 0.0f : 1.0f);
        R.memo_node = -1;
        C.node = -1;
        __syncthreads();
    } else {
        if (lane == 0) T.req[R.n_req] = (uint16_t)child;
        R.n_req += 1u;
        R.bytes += 4 + (G::HW + 2);
    }
}

// The K simulations of a round, BATCHED where the reference's sequence allows it (round 4).  Between two backups the descent reaches the same leaf (the memo), so the
// simulations of a round are: pick the r-th untried cell of that leaf (r from the simulation's own Philox block), place the stone, expand -- K times -- and only a terminal
// child (its backup changes n / w along the path) or a full leaf sends the next simulation down a new path.  run_sim does one simulation at a time (~550 instructions each, the
// wave's stream bound by their issue); here a leaf's simulations are taken together: lane i computes simulation i's Philox block, the picks are made in order (each shrinks
// the untried set: the sequential part, ~50 instructions per pick), the win checks follow pick by pick up to the first terminal child, and the children up to that one are
// written in ONE pass (lane i: table slot and header of child i; 2 NW lanes per child: its board words; requests in simulation order).  Node indices, insertion ranks, request
// order, statuses, the memo and the byte counter are those of the one-by-one sequence: the tests compare the trees bit for bit with the CPU restatement's.
template <int N>
__device__ void run_sims(const Store& S, const Tree<N>& T, Regs& R, LeafCache<N>& C, const RoundArgs& A, uint32_t first_sim, int count, uint32_t tree_global) {
    using G = Geo<N>;
    constexpr int ROWP = G::ROWP, NW = G::NW;
    const int lane = LANE;
    int done = 0;
#ifdef KROUND_PROF
    unsigned long long (&kp)[8] = R.kp;
    unsigned long long& kp_last = R.kp_last;
#endif
    PathRec rec;
    rec.depth = -1;
    NodeHdr root_h; // the root's header while nothing is added under the root (a round of terminal simulations starts at the root K times)
    bool root_ok = false;
    while (done < count) {
        // ---- the leaf (run_sim's head) ----
        int node = 0;
        uint32_t node_n = R.root_n;
        unsigned long long path_bytes = 0;
        if (R.memo_node >= 0) {
            node = R.memo_node;
            node_n = R.memo_n;
            path_bytes = R.memo_bytes;
        } else C.node = -1;
        NodeHdr h;
        const bool from_root = R.memo_node < 0; // (node == 0)
        if (C.node == node) h = C.h;
        else if (from_root && root_ok) h = root_h;
        else {
            fence_own_stores(R);
            h = T.hdr[node];
            if (from_root) { root_h = uni(h); root_ok = true; }
        }
        if (h.nch == h.legal && h.nch != 0) { fence_own_stores(R); C.node = -1; }
        if (from_root) select_leaf_impl<N, true>(T, h, node, node_n, path_bytes, rec);
        else {
            select_leaf<N>(T, h, node, node_n, path_bytes);
            rec.depth = -1;
        }
        KP(2); // descents (+ header loads)
        R.memo_node = node;
        R.memo_n = node_n;
        R.memo_bytes = path_bytes;
        if (h.status != ST_IN_PROGRESS) { // terminal leaf (pme.rs:92-97): one simulation
            R.bytes += path_bytes;
            if (rec.depth >= 0 && rec.depth <= PR_LEVELS) { // node.rs:83-99 along the recorded slots: the same additions, leaf's slot first, the sign alternating upwards
                float v = h.status >= ST_BLACK_WIN ? 1.0f : 0.0f;
#pragma unroll
                for (int l = PR_LEVELS - 1; l >= 0; --l)
                    if (l < rec.depth) {
                        if (lane == 0) {
                            const size_t slot = (size_t)rec.tab[l] * ROWP + rec.act[l];
                            T.cn[slot] = rec.n[l] + 1u;
                            T.cw[slot] = rec.w[l] + v;
                        }
                        v = -v;
                        R.bytes += 16;
                    }
                R.root_n += 1u;
                R.root_w += v;
                R.bytes += 16;
                R.dirty = true; // (the next descent reads these slots: it waits for the stores first)
            } else {
                fence_own_stores(R);
                backup<N>(T, R, node, h.status >= ST_BLACK_WIN ? 1.0f : 0.0f);
                __syncthreads();
            }
            R.memo_node = -1;
            C.node = -1;
            done += 1;
            KP(3); // terminal-leaf backups
            continue;
        }
        // ---- the leaf's board and untried cells (cached, or loaded once) ----
        uint64_t bb[2 * NW];
        unsigned long long cand[G::IT];
        int total = 0;
        if (C.node == node) {
#pragma unroll
            for (int i = 0; i < 2 * NW; ++i) bb[i] = C.bb[i];
#pragma unroll
            for (int j = 0; j < G::IT; ++j) { cand[j] = C.cand[j]; total += __popcll(cand[j]); }
        } else {
            fence_own_stores(R);
#pragma unroll
            for (int i = 0; i < 2 * NW; ++i) bb[i] = T.board[(size_t)node * (2 * NW) + i];
            uint8_t ordc[G::IT]; // (the insertion ranks go out with the board words, not behind them)
#pragma unroll
            for (int j = 0; j < G::IT; ++j) ordc[j] = h.table != NONE16 ? T.corder[(size_t)h.table * ROWP + j * 64 + lane] : NONE8;
#pragma unroll
            for (int j = 0; j < G::IT; ++j) {
                const int a = j * 64 + lane;
                const bool c = a < G::HW && !(((bb[j] | bb[NW + j]) >> lane) & 1ULL) && ordc[j] == NONE8;
                cand[j] = __ballot(c);
                total += __popcll(cand[j]);
            }
            C.node = node;
#pragma unroll
            for (int i = 0; i < 2 * NW; ++i) C.bb[i] = bb[i];
#pragma unroll
            for (int j = 0; j < G::IT; ++j) C.cand[j] = cand[j];
            C.h = h;
        }
        KP(4); // leaf board / untried cells
        const unsigned long long sim_bytes = path_bytes + 2 * (G::HW / 8);
        if (total == 0) { R.bytes += sim_bytes; done += 1; continue; } // "There's no action for now": this simulation is consumed
        // ---- how many simulations this leaf can take: the rest of the round, its untried cells, the room in the arenas ----
        int m = count - done;
        if (m > total) m = total;
        const int room = S.cap_nodes - (int)R.n_nodes;
        if (m > room) m = room > 0 ? room : 0;
        if (h.table == NONE16 && R.n_tables >= (uint32_t)S.cap_tables) m = 0;
        if (m == 0) { // arena overflow: this and every later simulation of the round fails in expand, as one by one
            R.error |= 1u;
            R.bytes += sim_bytes * (unsigned long long)(count - done);
            C.node = -1;
            return;
        }
        // ---- the picks, in simulation order (pme.rs:101-125): simulation done + i takes the r-th cell of what the earlier ones left ----
        uint32_t my_x = 0u;
        if (lane < m) my_x = philox(A.seed, first_sim + (uint32_t)(done + lane), (uint32_t)A.ply, tree_global, RNG_EXPAND).x;
        int my_action = 0, my_status = ST_IN_PROGRESS;
#if PARALLEL_PICKS
        {
            // Simulation i draws r_i = floor(x_i (total - i) / 2^32) and takes the r_i-th of the cells the earlier ones left.  In RANK space (rank = position among the leaf's
            // untried cells as they are now): with the ranks taken so far sorted, s_0 < s_1 < ..., the r-th remaining rank is r + #{j : s_j - j <= r}.  The sorted list lives across
            // the lanes (lane j: s_j), so a pick is one ballot and one shuffle; the ranks turn into cells for all simulations at once.  The same cells as picking one by one
            // (the loop this replaces: ~1100 cycles per pick, a quarter of the kernel for the average tree).
            const int my_r = lane < m ? (int)__umulhi(my_x, (uint32_t)(total - lane)) : 0;
            int sorted = 0x7FFFFFFF, my_q = 0; // lane j: the j-th smallest rank taken so far
            for (int i = 0; i < m; ++i) {
                const int r = __builtin_amdgcn_readlane(my_r, i);
                const int k = __popcll(__ballot(lane < i && sorted - lane <= r));
                const int q = r + k;
                const int up = __shfl_up(sorted, 1, 64);
                sorted = lane < k ? sorted : (lane == k ? q : up);
                if (lane == i) my_q = q;
            }
            // rank -> cell: the my_q-th set bit of the untried mask
            int cum = 0, wj = 0, rr = my_q;
            uint64_t wsel = cand[0];
#pragma unroll
            for (int j = 0; j < G::IT; ++j) {
                const int c = __popcll(cand[j]);
                if (my_q >= cum && my_q < cum + c) { wj = j; rr = my_q - cum; wsel = cand[j]; }
                cum += c;
            }
            if (lane < m) my_action = wj * 64 + nth_set_bit_lane(wsel, rr);
            for (int i = 0; i < m; ++i) { // the picked cells are no longer untried
                const int a = __builtin_amdgcn_readlane(my_action, i);
#pragma unroll
                for (int j = 0; j < G::IT; ++j) cand[j] &= ~((a >> 6) == j ? (1ULL << (a & 63)) : 0ULL);
            }
        }
#else
        {
            int tot = total;
            for (int i = 0; i < m; ++i) {
                const uint32_t x = (uint32_t)__builtin_amdgcn_readlane((int)my_x, i);
                int r = (int)__umulhi(x, (uint32_t)tot);
                int action = 0;
                bool found = false;
#pragma unroll
                for (int j = 0; j < G::IT; ++j) {
                    const int c = __popcll(cand[j]);
                    if (!found) {
                        if (r < c) { action = j * 64 + nth_set_bit(cand[j], r); found = true; }
                        else r -= c;
                    }
                }
#pragma unroll
                for (int j = 0; j < G::IT; ++j) cand[j] &= ~((action >> 6) == j ? (1ULL << (action & 63)) : 0ULL);
                if (lane == i) my_action = action;
                tot -= 1;
            }
        }
#endif
        KP(5); // picks
        // ---- place the stones (pme.rs:128-135): win check per child, up to the first terminal one ----
        int n_commit = m, term_status = ST_IN_PROGRESS;
#if PARALLEL_PICKS
        {
            // lane 4 c + p checks direction pair p of child c (16 children at a time); a child is a win if any of its four pairs makes exactly five, a draw if it took the
            // leaf's last cell; the first terminal child ends the batch (the loop this replaces checked one child at a time with 40 lanes: ~1300 cycles each)
            uint64_t own[NW];
#pragma unroll
            for (int w = 0; w < NW; ++w) own[w] = h.turn == 0 ? bb[w] : bb[NW + w];
            int first = -1;
            bool first_five = false;
            for (int c0 = 0; c0 < m && first < 0; c0 += 16) {
                const int c = c0 + (lane >> 2);
                const int a = __shfl(my_action, c < m ? c : 0, 64);
                const bool five = c < m && five_in_pair<N>(own, a, lane & 3);
                unsigned long long fm = __ballot(five);
                fm |= fm >> 1;
                fm |= fm >> 2; // bit 4 k: child c0 + k has a five
                fm &= 0x1111111111111111ULL;
                if (fm) { first = c0 + ((__ffsll((long long)fm) - 1) >> 2); first_five = true; }
            }
            if (h.legal == 1) { first_five = first == 0; first = 0; } // (m == 1: the child fills the board)
            if (first >= 0) {
                n_commit = first + 1;
                term_status = first_five ? (h.turn == 0 ? ST_BLACK_WIN : ST_WHITE_WIN) : ST_DRAW;
                if (lane == first) my_status = term_status;
            }
        }
#else
        for (int i = 0; i < m; ++i) {
            const int a = __builtin_amdgcn_readlane(my_action, i);
            uint64_t mine[NW];
#pragma unroll
            for (int w = 0; w < NW; ++w) mine[w] = h.turn == 0 ? bb[w] : bb[NW + w];
            set_bit<NW>(mine, a);
            const int five = exactly_five<N>(mine, a);
            const int st = five ? (h.turn == 0 ? ST_BLACK_WIN : ST_WHITE_WIN) : (h.legal == 1 ? ST_DRAW : ST_IN_PROGRESS);
            if (lane == i) my_status = st;
            if (st != ST_IN_PROGRESS) { n_commit = i + 1; term_status = st; break; }
        }
#endif
        KP(6); // win checks
        // ---- expand (node.rs:61-81), n_commit children in one pass ----
        int tab = h.table;
        if (tab == NONE16) {
            tab = (int)R.n_tables++;
#pragma unroll
            for (int j = 0; j < G::IT; ++j) T.corder[(size_t)tab * ROWP + j * 64 + lane] = NONE8;
            if (lane == 0) { T.owner[tab] = (uint16_t)node; T.hdr[node].table = (uint16_t)tab; }
            __syncthreads(); // corder row init must land before the slot stores below
        }
        const int idx0 = (int)R.n_nodes;
        R.n_nodes += (uint32_t)n_commit;
        if (lane < n_commit) {
            const size_t slot = (size_t)tab * ROWP + my_action;
            T.corder[slot] = (uint8_t)(h.nch + lane);
            T.cidx[slot] = (uint16_t)(idx0 + lane);
            T.cn[slot] = 0u;
            T.cw[slot] = 0.0f;
            NodeHdr c;
            c.parent = (uint16_t)node;
            c.table = NONE16;
            c.legal = (uint16_t)(h.legal - 1);
            c.nch = 0;
            c.action = (uint8_t)my_action;
            c.status = (uint8_t)my_status;
            c.turn = (uint8_t)(1 - h.turn);
            c.has_policy = 0;
            c.pad = 0;
            T.hdr[idx0 + lane] = c;
        }
        if (lane == 0) T.hdr[node].nch = (uint16_t)(h.nch + n_commit);
        for (int t0 = 0; t0 < n_commit * 2 * NW; t0 += 64) { // the children's boards: the leaf's words + the new stone
            const int t = t0 + lane, ci = t / (2 * NW), wi = t % (2 * NW);
            const int a = __shfl(my_action, ci < n_commit ? ci : 0, 64);
            if (t < n_commit * 2 * NW) {
                uint64_t w = bb[0];
#pragma unroll
                for (int k = 1; k < 2 * NW; ++k) w = wi == k ? bb[k] : w;
                const bool own = (wi < NW) == (h.turn == 0);
                if (own && (a >> 6) == (wi % NW)) w |= 1ULL << (a & 63);
                T.board[(size_t)(idx0 + ci) * (2 * NW) + wi] = w;
            }
        }
        const int n_new_req = n_commit - (term_status != ST_IN_PROGRESS ? 1 : 0); // (only the last committed child can be terminal)
        if (lane < n_new_req) T.req[R.n_req + lane] = (uint16_t)(idx0 + lane);
        R.n_req += (uint32_t)n_new_req;
        R.bytes += (sim_bytes + 24 + 8 * 2 * NW) * (unsigned long long)n_commit + (unsigned long long)(4 + (G::HW + 2)) * (unsigned long long)n_new_req;
        R.dirty = true;
        h.table = (uint16_t)tab;
        h.nch = (uint16_t)(h.nch + n_commit);
        if (node == 0) root_ok = false; // (the root's stored header has changed)
        if (term_status != ST_IN_PROGRESS) { // pme.rs:177-181: the terminal child's reward goes up the path now; the next simulation starts at the root again
            fence_own_stores(R);
            backup<N>(T, R, idx0 + n_commit - 1, term_status == ST_DRAW ? 0.0f : 1.0f);
            R.memo_node = -1;
            C.node = -1;
            __syncthreads();
        } else { // the leaf as it is now (no terminal child: every pick was committed)
            C.h = h;
#pragma unroll
            for (int j = 0; j < G::IT; ++j) C.cand[j] = cand[j];
        }
        KP(7); // expansion stores (+ a terminal child's backup)
        done += n_commit;
    }
}

#ifndef SCATTER_SEGMENTS
#define SCATTER_SEGMENTS 1 // (A-B builds: 0 = rounds with several request parents back up one request at a time, as in round 4; same results)
#endif
#ifndef KROUND_BATCH
#define KROUND_BATCH 1 // 0: one simulation at a time (run_sim; A-B builds and the reference point of the batched form)
#endif
#ifndef KROUND_WPS
#define KROUND_WPS 4 // minimum waves per SIMD the register allocation of k_round must allow (A-B builds: -DKROUND_WPS=...)
#endif
template <int N>
__global__ __launch_bounds__(64, KROUND_WPS) void k_round(Store S, RoundArgs A) {
    using G = Geo<N>;
    __shared__ float s_row[G::ROWP];
    const int g = blockIdx.x;
    const int t = A.side * S.games + g;
    const Tree<N> T(S, t);
    const GameState gs0 = S.gs[g];
    const TreeState ts = *T.ts; // (issued with the game's state, not behind it)
    if (!gs0.alive) return;
    Regs R{ts.n_nodes, ts.n_tables, ts.root_n, 0u, ts.error, ts.root_w, 0ull};
#ifdef KROUND_PROF
    unsigned long long (&kp)[8] = R.kp;
    unsigned long long& kp_last = R.kp_last;
    kp_last = __builtin_readcyclecounter();
#endif
#ifndef KROUND_EXP
#define KROUND_EXP 0 // timing-only A-B builds: 1 = no deferred backups, 2 = no simulations
#endif
#if KROUND_EXP != 0 && !defined(OMOK_EXPERIMENT)
#error "KROUND_EXP builds are timing experiments with wrong results: build them with -DOMOK_EXPERIMENT, never as the product"
#endif
    if (KROUND_EXP != 1 && A.scatter_v && ts.n_req > 0) { // the previous round's backups, deferred into this kernel (run loop)
        scatter_tree<N>(S, T, ts, R, A.scatter_v);
        R.dirty = true;
    }
    // RNG streams are keyed by the GAME (its global id and its own ply), not by the slot or the engine's ply counter: in an episode the
    // two coincide; in slots mode a slot's later games have their own ids and start their plies at 0
    A.ply = gs0.plies;
    const uint32_t tree_global = (uint32_t)((A.game_offset + gs0.gid) * 2 + A.side);
    KP(0); // state loads + the previous round's backups
    if (A.round == 0) apply_noise<N>(T, A, tree_global, s_row);
    KP(1); // noise
    LeafCache<N> C;
    if (KROUND_EXP == 2) { /* timing only */ }
    else if (KROUND_BATCH) run_sims<N>(S, T, R, C, A, (uint32_t)(A.round * A.K), A.K, tree_global);
    else for (int i = 0; i < A.K; ++i) run_sim<N>(S, T, R, C, A, (uint32_t)(A.round * A.K + i), tree_global);
    if (LANE == 0) {
        TreeState o = ts;
        o.n_nodes = R.n_nodes; o.n_tables = R.n_tables; o.root_n = R.root_n; o.root_w = R.root_w;
        o.error = R.error; o.n_req = R.n_req;
        *T.ts = o;
        atomicAdd(S.d_bytes, R.bytes);
#ifdef KROUND_PROF
        const unsigned int L = (unsigned int)A.round & 63u;
        if (t < 8192)
            for (int i = 0; i < 8; ++i) g_kprof[L][t][i] = kp[i];
#endif
    }
}
#ifdef KROUND_PROF
extern "C" void omok_debug_kround_prof(unsigned long long* out /* [64][8192][8] */) {
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(out, HIP_SYMBOL(g_kprof), sizeof(unsigned long long) * 64 * 8192 * 8);
    static unsigned long long zero[1] = {0};
    (void)zero;
}
#endif

// ---------------------------------------------------------------------------------------------
// ONE TREE SEARCHED BY MANY WAVES: MCTSExecutor::run (alpha-zero/src/mcts_executor.rs:29-255), the executor of the GUI and of
// the checkpoint-vs-checkpoint match.  The reference runs ceil(count / batch_size) rounds of `batch_size` simulations as rayon
// tasks on ONE shared tree: relaxed atomics on n / w (node.rs:86-88), the children lock in expand() -- a second thread that
// picked the same action gets None and drops its simulation (mcts_executor.rs:171-178) -- and its own evaluate_pv per round.
// Here: W waves of ONE workgroup (one CU: its waves share the vector L1, so workgroup-scope fences order their global
// accesses) run W rounds concurrently; their requests are evaluated as one batch and scattered by the same W waves.  That is
// one of the schedules the reference's thread pool can produce (W threads that reach evaluate_pv together), and with W = 1 it
// is the sequential schedule of omok_execute, bit for bit.  Results for W > 1 depend on the interleaving, as in the reference.
// ---------------------------------------------------------------------------------------------
__device__ inline void wave_fence() { // orders this wave's global accesses against the other waves of the workgroup
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_wave_barrier();
}
__device__ inline void tree_lock(int* lock) { // the allocator Mutex + children write lock of mcts/src/lib.rs:43-45, node.rs:67
    if (LANE == 0)
        while (atomicCAS(lock, 0, 1) != 0) __builtin_amdgcn_s_sleep(2);
    wave_fence();
}
__device__ inline void tree_unlock(int* lock) {
    wave_fence();
    if (LANE == 0) atomicExch(lock, 0);
}

template <int N>
__device__ inline void backup_shared(const Tree<N>& T, int x, float v) { // node.rs:83-99 with fetch_add (n first, then w)
    constexpr int ROWP = Geo<N>::ROWP;
    const volatile NodeHdr* hdr = T.hdr;
    while (x != 0) {
        const uint16_t parent = hdr[x].parent;
        const size_t slot = (size_t)hdr[parent].table * ROWP + hdr[x].action;
        if (LANE == 0) { atomicAdd(&T.cn[slot], 1u); atomicAdd(&T.cw[slot], v); }
        v = -v;
        x = parent;
    }
    if (LANE == 0) { atomicAdd(&T.ts->root_n, 1u); atomicAdd(&T.ts->root_w, v); }
}

// Node::expand under the tree lock; returns the node index, -1 on arena overflow, -2 if another wave expanded the action first
template <int N>
__device__ inline int add_child_shared(const Store& S, const Tree<N>& T, int* lock, int parent, int action, const uint64_t* bb,
                                       int status, int turn, bool lock_held = false) {
    using G = Geo<N>;
    const int lane = LANE;
    volatile NodeHdr* hdr = T.hdr;
    volatile TreeState* ts = T.ts;
    volatile uint8_t* corder = T.corder;
    if (!lock_held) tree_lock(lock);
    int result;
    const uint16_t nch = hdr[parent].nch, legal = hdr[parent].legal;
    int tab = hdr[parent].table;
    const uint32_t n_nodes = ts->n_nodes, n_tables = ts->n_tables;
    if (tab != NONE16 && corder[(size_t)tab * G::ROWP + action] != NONE8) {
        result = -2; // children.iter().any(|child| child.action == Some(action)) -> None (node.rs:69-71)
    } else if (n_nodes >= (uint32_t)S.cap_nodes || (tab == NONE16 && n_tables >= (uint32_t)S.cap_tables)) {
        if (lane == 0) ts->error = ts->error | 1u;
        result = -1;
    } else {
        if (tab == NONE16) {
            tab = (int)n_tables;
#pragma unroll
            for (int j = 0; j < G::IT; ++j) corder[(size_t)tab * G::ROWP + j * 64 + lane] = NONE8;
            if (lane == 0) { T.owner[tab] = (uint16_t)parent; hdr[parent].table = (uint16_t)tab; ts->n_tables = n_tables + 1u; }
        }
        const int idx = (int)n_nodes;
        const size_t slot = (size_t)tab * G::ROWP + action;
        if (lane == 0) {
            NodeHdr c;
            c.parent = (uint16_t)parent; c.table = NONE16; c.legal = (uint16_t)(legal - 1); c.nch = 0;
            c.action = (uint8_t)action; c.status = (uint8_t)status; c.turn = (uint8_t)turn; c.has_policy = 0; c.pad = 0;
            T.hdr[idx] = c;
            T.cidx[slot] = (uint16_t)idx;
            T.cn[slot] = 0u;
            T.cw[slot] = 0.0f;
        }
        if (lane < 2 * G::NW) {
            uint64_t w = bb[0];
#pragma unroll
            for (int i = 1; i < 2 * G::NW; ++i) w = lane == i ? bb[i] : w;
            T.board[(size_t)idx * (2 * G::NW) + lane] = w;
        }
        wave_fence(); // the child is complete before it becomes reachable
        if (lane == 0) { corder[slot] = (uint8_t)nch; hdr[parent].nch = (uint16_t)(nch + 1); ts->n_nodes = n_nodes + 1u; }
        result = idx;
    }
    if (!lock_held) tree_unlock(lock);
    return result;
}

// RECORDED mode (rec_order != NULL; omok_execute_shared_recorded): a wave holds the tree lock for its WHOLE simulation -- descent,
// expansion and a terminal backup -- and writes its index into rec_order at the position the lock order gives it.  The run is then a
// sequential interleaving of the waves' simulations (one of the schedules the reference's pool can produce) that a CPU restatement can
// replay exactly (tests/test_gpu_shared_tree.py), so trees of W > 1 searches compare bit for bit instead of statistically.
template <int N>
__device__ void run_sim_shared_body(const Store& S, const Tree<N>& T, int* lock, const RoundArgs& A, uint32_t sim_index, uint32_t tree_global,
                                    uint16_t* my_req, uint32_t& n_req, bool lock_held);
__device__ inline void record_turn(uint8_t* rec_order, uint32_t* rec_pos, int wave) { // (under the tree lock)
    if (LANE == 0) {
        volatile uint32_t* pos = rec_pos;
        const uint32_t p = *pos;
        ((volatile uint8_t*)rec_order)[p] = (uint8_t)wave;
        *pos = p + 1u;
    }
}
template <int N>
__device__ void run_sim_shared(const Store& S, const Tree<N>& T, int* lock, const RoundArgs& A, uint32_t sim_index, uint32_t tree_global,
                               uint16_t* my_req, uint32_t& n_req, uint8_t* rec_order, uint32_t* rec_pos, int wave) {
    if (rec_order) {
        tree_lock(lock);
        record_turn(rec_order, rec_pos, wave);
    }
    run_sim_shared_body<N>(S, T, lock, A, sim_index, tree_global, my_req, n_req, rec_order != nullptr);
    if (rec_order) tree_unlock(lock);
}
template <int N>
__device__ void run_sim_shared_body(const Store& S, const Tree<N>& T, int* lock, const RoundArgs& A, uint32_t sim_index, uint32_t tree_global,
                                    uint16_t* my_req, uint32_t& n_req, bool lock_held) {
    using G = Geo<N>;
    constexpr int ROWP = G::ROWP, NW = G::NW;
    const int lane = LANE;
    const volatile NodeHdr* hdr = T.hdr;
    const volatile uint32_t* cn = T.cn;
    const volatile float* cw = T.cw;
    const volatile uint8_t* corder = T.corder;
    const volatile uint16_t* cidx = T.cidx;
    int node = 0;
    uint32_t node_n = ((const volatile TreeState*)T.ts)->root_n;
    uint16_t h_nch = hdr[0].nch, h_legal = hdr[0].legal, h_table = hdr[0].table;
    uint8_t h_has = hdr[0].has_policy;
    // ---- select_leaf (node.rs:43-58), PUCT on whatever n / w the other waves have published so far ----
    while (h_nch == h_legal && h_nch != 0) {
        const uint32_t pn = node_n > 1u ? node_n : 1u;
        const float sq = __fsqrt_rn((float)pn);
        const size_t tb = (size_t)h_table * ROWP;
        const float ph = __fdiv_rn(1.0f, (float)h_legal);
        unsigned long long best = 0ULL;
#pragma unroll
        for (int j = 0; j < G::IT; ++j) {
            const int a = j * 64 + lane;
            const uint8_t ord = corder[tb + a];
            if (ord != NONE8) {
                const uint32_t n = cn[tb + a];
                const float w = cw[tb + a];
                const float p = h_has ? T.pol[(size_t)node * ROWP + a] : ph;
                const float q = __fdiv_rn(w, (float)n + F32_EPS);
                const float bias = __fdiv_rn(sq, (float)(1u + n));
                const float score = q + (1.0f * p) * bias;
                const unsigned long long v = ((unsigned long long)total_key_biased(score) << 32) | ((unsigned long long)ord << 16) | (unsigned long long)a;
                best = v > best ? v : best;
            }
        }
        best = wave_max_u64(best);
        const int a_best = (int)(best & 0xFFFFu);
        node_n = cn[tb + a_best];
        node = (int)cidx[tb + a_best];
        h_nch = hdr[node].nch; h_legal = hdr[node].legal; h_table = hdr[node].table; h_has = hdr[node].has_policy;
    }
    const int h_status = hdr[node].status, h_turn = hdr[node].turn;
    if (h_status != ST_IN_PROGRESS) { // pme / mcts_executor.rs:92-97
        backup_shared<N>(T, node, h_status >= ST_BLACK_WIN ? 1.0f : 0.0f);
        return;
    }
    uint64_t bb[2 * NW];
#pragma unroll
    for (int i = 0; i < 2 * NW; ++i) bb[i] = T.board[(size_t)node * (2 * NW) + i]; // a node's board never changes once published
    unsigned long long cand[G::IT];
    int total = 0;
#pragma unroll
    for (int j = 0; j < G::IT; ++j) {
        const int a = j * 64 + lane;
        bool c = a < G::HW && !(((bb[j] | bb[NW + j]) >> lane) & 1ULL);
        if (c && h_table != NONE16) c = corder[(size_t)h_table * ROWP + a] == NONE8;
        cand[j] = __ballot(c);
        total += __popcll(cand[j]);
    }
    if (total == 0) return;
    const U4 o = philox(A.seed, sim_index, (uint32_t)A.ply, tree_global, RNG_EXPAND);
    int r = (int)__umulhi(o.x, (uint32_t)total);
    int action = 0;
    bool found = false;
#pragma unroll
    for (int j = 0; j < G::IT; ++j) {
        const int c = __popcll(cand[j]);
        if (!found) {
            if (r < c) { action = j * 64 + nth_set_bit(cand[j], r); found = true; }
            else r -= c;
        }
    }
    const int status = place_and_status<N>(bb, h_turn, h_legal, action);
    const int child = add_child_shared<N>(S, T, lock, node, action, bb, status, 1 - h_turn, lock_held);
    if (child < 0) return; // None: "already expanded by other thread" (mcts_executor.rs:171-178), or the arena is full
    if (status != ST_IN_PROGRESS) backup_shared<N>(T, child, status == ST_DRAW ? 0.0f : 1.0f);
    else { if (lane == 0) my_req[n_req] = (uint16_t)child; n_req += 1u; }
}

// W waves (one workgroup) x one round each: wave w runs round `group * W + w` of the execute call
template <int N>
__global__ __launch_bounds__(1024) void k_round_shared(Store S, RoundArgs A, int rounds_total, int group, uint16_t* __restrict__ sh_req,
                                                       uint32_t* __restrict__ sh_cnt, uint8_t* rec_order, uint32_t* rec_pos) {
    __shared__ int s_lock;
    if (threadIdx.x == 0) s_lock = 0;
    __syncthreads();
    const int wave = threadIdx.x >> 6, W = blockDim.x >> 6;
    const int t = A.side * S.games; // game 0
    const Tree<N> T(S, t);
    const uint32_t tree_global = (uint32_t)(A.game_offset * 2 + A.side);
    const int round = group * W + wave;
    uint32_t n_req = 0;
    if (S.gs[0].alive && round < rounds_total)
        for (int i = 0; i < A.K; ++i)
            run_sim_shared<N>(S, T, &s_lock, A, (uint32_t)(round * A.K + i), tree_global, sh_req + (size_t)wave * KMAX, n_req, rec_order, rec_pos, wave);
    if (LANE == 0) sh_cnt[wave] = n_req;
}

// request list of the W rounds in round order, then simulation order (one wave)
__global__ __launch_bounds__(64) void k_scan_shared(Store S, int side, int W, const uint16_t* __restrict__ sh_req, uint32_t* __restrict__ sh_cnt) {
    const int lane = threadIdx.x;
    uint32_t base = 0;
    for (int w = 0; w < W; ++w) {
        const uint32_t c = sh_cnt[w];
        if ((uint32_t)lane < c) {
            S.req_ref[base + lane] = ((uint32_t)(side * S.games) << 16) | sh_req[(size_t)w * KMAX + lane];
            S.req_aux[base + lane] = 0xFFFFFFFFu;
        }
        if (lane == 0) sh_cnt[KMAX + w] = base; // first request of round w
        base += c;
    }
    if (lane == 0) S.d_count[0] = (int32_t)base;
}

// the scatter halves of the W rounds (mcts_executor.rs:206-250): every wave backs its own requests up, fetch_add like propagate
template <int N>
__global__ __launch_bounds__(1024) void k_scatter_shared(Store S, int side, const float* __restrict__ V, const uint16_t* __restrict__ sh_req,
                                                         const uint32_t* __restrict__ sh_cnt, uint8_t* rec_order, uint32_t* rec_pos) {
    __shared__ int s_lock;
    if (threadIdx.x == 0) s_lock = 0;
    __syncthreads();
    const int wave = threadIdx.x >> 6;
    const Tree<N> T(S, side * S.games);
    const uint32_t c = sh_cnt[wave], base = sh_cnt[KMAX + wave];
    for (uint32_t r = 0; r < c; ++r) {
        if (rec_order) { // recorded mode: one backup at a time, in the recorded order (float sums depend on it)
            tree_lock(&s_lock);
            record_turn(rec_order, rec_pos, wave);
        }
        backup_shared<N>(T, sh_req[(size_t)wave * KMAX + r], -V[base + r]);
        if (rec_order) tree_unlock(&s_lock);
    }
}

// ---------------------------------------------------------------------------------------------
// k_scan: dense, order-preserving request list over the live trees of one side
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_scan(Store S, int side, unsigned long long* __restrict__ evals, int32_t* __restrict__ zero_ptr, int zero_n) {
    // (zero_ptr: the counters of the net's sibling grouping, which the round's forward expects zeroed: one launch less per round)
    for (int i = threadIdx.x; i < zero_n; i += 1024) zero_ptr[i] = 0;
    // exclusive scan of the per-tree request counts in game order: each thread owns a contiguous chunk of
    // games, chunk sums are scanned with wave shuffles (64 lanes) and a 16-entry LDS table
    __shared__ uint32_t s_wave[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    static_assert(MAX_GAMES <= SCAN_GAMES_PER_THREAD * 1024, "k_scan covers 1024 threads x SCAN_GAMES_PER_THREAD games");
    const int chunk = (S.games + 1023) / 1024; // <= SCAN_GAMES_PER_THREAD (omok_create: games <= MAX_GAMES)
    const int g0 = tid * chunk;
    uint32_t local = 0;
    // (both loads of every game issued unconditionally and kept: `alive` then `n_req` as two dependent round trips per game and a second
    //  pass over the same pairs cost 60 us at 16384 games)
    uint32_t cnt[SCAN_GAMES_PER_THREAD];
#pragma unroll
    for (int i = 0; i < SCAN_GAMES_PER_THREAD; ++i) {
        const int g = g0 + i;
        const bool in = i < chunk && g < S.games;
        const uint32_t alive = in ? (uint32_t)S.gs[in ? g : 0].alive : 0u;
        const uint32_t nreq = in ? S.ts[side * S.games + (in ? g : 0)].n_req : 0u;
        cnt[i] = alive ? nreq : 0xFFFFFFFFu; // (0xFFFFFFFF: no live game here)
        local += alive ? nreq : 0u;
    }
    uint32_t incl = local;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t v = __shfl_up(incl, o, 64);
        if (lane >= o) incl += v;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint32_t wbase = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
        const uint32_t v = s_wave[w];
        if (w < wave) wbase += v;
        total += v;
    }
    uint32_t run = wbase + incl - local;
#pragma unroll
    for (int i = 0; i < SCAN_GAMES_PER_THREAD; ++i) {
        if (cnt[i] != 0xFFFFFFFFu) {
            S.ts[side * S.games + g0 + i].req_base = run;
            run += cnt[i];
        }
    }
    if (tid == 0) {
        S.d_count[0] = (int32_t)total;
        if (evals) evals[0] += total; // (the evaluation counter of the stats: one launch less per round than a kernel of its own)
    }
}

// dense request list (tree, node) in tree order then simulation order (pme.rs:194-205)
__global__ __launch_bounds__(256) void k_fill(Store S, int side, int K) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int g = i / K, r = i % K;
    if (g >= S.games || !S.gs[g].alive) return;
    const int t = side * S.games + g;
    const TreeState ts = S.ts[t];
    if ((uint32_t)r >= ts.n_req) return;
    S.req_ref[ts.req_base + r] = ((uint32_t)t << 16) | S.req_node[(size_t)t * KMAX + r];
    S.req_aux[ts.req_base + r] = 0xFFFFFFFFu;
}

// ---------------------------------------------------------------------------------------------
// k_scatter (pme.rs:222-265): requests of a tree applied in simulation order
// ---------------------------------------------------------------------------------------------
// part 1, one wave per REQUEST (requests are independent here): mask the occupied cells, renormalise with
// the sequential f32 sum of pme.rs:241-249, overwrite the node's policy row (pme.rs:235-252)
template <int N>
__global__ __launch_bounds__(64) void k_scatter_policy(Store S, const float* __restrict__ P, int max_count) {
    using G = Geo<N>;
    constexpr int ROWP = G::ROWP, NW = G::NW;
    int count = S.d_count[0];
    if (count > max_count) count = max_count;
    const int lane = LANE;
    // (four requests per iteration, their loads in flight together, measured SLOWER: 100 vs 67 us per 65536 requests)
    for (int d = blockIdx.x; d < count; d += gridDim.x) {
        const uint32_t ref = S.req_ref[d];
        const size_t tn = (size_t)(ref >> 16) * (size_t)S.stride_nodes + (size_t)(ref & 0xFFFFu);
        uint64_t occ[NW];
#pragma unroll
        for (int i = 0; i < NW; ++i) occ[i] = S.board[tn * (2 * NW) + i] | S.board[tn * (2 * NW) + NW + i];
        float row[G::IT];
#pragma unroll
        for (int j = 0; j < G::IT; ++j) {
            const int a = j * 64 + lane;
            const bool empty = a < G::HW && !((occ[j] >> lane) & 1ULL);
            row[j] = empty ? P[(size_t)d * ROWP + a] : 0.0f; // pme.rs:235-239
        }
        const float sum = seq_sum_regs<N>(row);
        const bool renorm = F32_EPS <= sum;
        const float inv = renorm ? __fdiv_rn(1.0f, sum) : 1.0f;
#pragma unroll
        for (int j = 0; j < G::IT; ++j) S.policy[tn * ROWP + j * 64 + lane] = renorm ? row[j] * inv : row[j];
        if (lane == 0) S.hdr[tn].has_policy = 1;
    }
}

// k_softmax (net_kernels.hip) + part 1 in one pass for the rounds of the run loop: the probabilities go from registers straight into the
// node's policy row instead of through a [requests][ROWP] array in HBM and a second kernel.  The softmax is k_softmax's, operation for
// operation (the step-wise API evaluates with the separate kernels and must produce the same bits), the rest is part 1's.
// Round 4: the sequential f32 sum of pme.rs:241-249 (225 dependent additions in cell order: `iter().sum()`) was 450 of the ~600 vector instructions
// a request cost here (v_readlane + v_add per cell, one request per wave: the kernel was bound by their issue slots, 87 us per 65536 requests).  A wave now
// takes SSP_BATCH requests: their masked rows go to LDS [request][cell], lane r then adds up request r's row IN THE SAME ORDER (one chain of 225 additions
// serves the whole batch), and the rows are read back, scaled and stored.  HW is odd, so the lanes of the summing phase hit different banks.
#ifndef SSP_BATCH_N
#define SSP_BATCH_N 4 // (per 65536 requests: 4 -> 46.9 us, 8 -> 50.0, 16 -> 80.1: the LDS rows limit the waves per CU; A-B builds: -DSSP_BATCH_N=...)
#endif
constexpr int SSP_BATCH = SSP_BATCH_N;
template <int N>
__global__ __launch_bounds__(64) void k_softmax_scatter_policy(Store S, const float* __restrict__ logits, int lrow, float* __restrict__ V,
                                                               float* __restrict__ Vpre, int max_count) {
    using G = Geo<N>;
    constexpr int ROWP = G::ROWP, NW = G::NW, hw = G::HW;
    static_assert((hw & 1) == 1, "odd row stride: conflict-free column reads");
    __shared__ float srow[SSP_BATCH * hw];
    __shared__ float sinv[SSP_BATCH];
    int count = S.d_count[0];
    if (count > max_count) count = max_count;
    const int lane = LANE;
    const int nb = (count + SSP_BATCH - 1) / SSP_BATCH;
    for (int b = blockIdx.x; b < nb; b += gridDim.x) {
        const int d0 = b * SSP_BATCH;
        const int nreq = count - d0 < SSP_BATCH ? count - d0 : SSP_BATCH; // (wave-uniform)
        // ---- every load of the batch up front: request refs (lane i = request i), logits, occupancy words of the nodes ----
        const uint32_t my_ref = lane < nreq ? S.req_ref[d0 + lane] : 0u;
        float lv[SSP_BATCH][G::IT];
#pragma unroll
        for (int i = 0; i < SSP_BATCH; ++i) {
            const float* l = logits + (size_t)(d0 + (i < nreq ? i : 0)) * lrow;
#pragma unroll
            for (int jj = 0; jj < G::IT; ++jj) lv[i][jj] = lane + 64 * jj < hw ? l[lane + 64 * jj] : -INFINITY;
        }
        const float my_vpre = lane < nreq ? logits[(size_t)(d0 + lane) * lrow + hw] : 0.0f;
        size_t tn[SSP_BATCH];
        uint64_t occ[SSP_BATCH][NW];
#pragma unroll
        for (int i = 0; i < SSP_BATCH; ++i) {
            const uint32_t ref = (uint32_t)__builtin_amdgcn_readlane((int)my_ref, i);
            tn[i] = (size_t)(ref >> 16) * (size_t)S.stride_nodes + (size_t)(ref & 0xFFFFu);
#pragma unroll
            for (int w = 0; w < NW; ++w) occ[i][w] = i < nreq ? (S.board[tn[i] * (2 * NW) + w] | S.board[tn[i] * (2 * NW) + NW + w]) : 0ULL;
        }
        if (lane < nreq) { V[d0 + lane] = tanhf(my_vpre); Vpre[d0 + lane] = my_vpre; }
        // ---- phase 1: k_softmax's arithmetic per request, the masked row -> LDS ----
#pragma unroll
        for (int i = 0; i < SSP_BATCH; ++i) {
            if (i < nreq) {
                float mx = -INFINITY;
#pragma unroll
                for (int jj = 0; jj < G::IT; ++jj) mx = fmaxf(mx, lv[i][jj]); // (ascending cells, as k_softmax's loop)
                for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
                float sum = 0.0f, row[G::IT];
#pragma unroll
                for (int jj = 0; jj < G::IT; ++jj) {
                    const int a = lane + 64 * jj;
                    row[jj] = 0.0f;
                    if (a < hw) { row[jj] = expf(lv[i][jj] - mx); sum += row[jj]; }
                }
                for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
#pragma unroll
                for (int jj = 0; jj < G::IT; ++jj) {
                    const int a = jj * 64 + lane;
                    const float p = a < hw ? row[jj] / sum : 0.0f;       // the softmax output of this cell
                    const bool empty = a < hw && !((occ[i][jj] >> lane) & 1ULL);
                    if (a < hw) srow[i * hw + a] = empty ? p : 0.0f;     // pme.rs:235-239
                }
            }
        }
        __syncthreads();
        // ---- phase 2: lane r = request r: the sequential sum over its row, cells ascending (pme.rs:241-249) ----
        if (lane < nreq) {
            float acc = 0.0f;
#pragma unroll 15
            for (int a = 0; a < hw; ++a) acc += srow[lane * hw + a];
            const bool renorm = F32_EPS <= acc;
            sinv[lane] = renorm ? __fdiv_rn(1.0f, acc) : 0.0f; // (0 = no renormalisation: a kept sum is >= EPS, its reciprocal > 0)
        }
        __syncthreads();
        // ---- phase 3: rows back from LDS, scaled, into the nodes' policy rows ----
#pragma unroll
        for (int i = 0; i < SSP_BATCH; ++i) {
            if (i < nreq) {
                const float inv = sinv[i];
                const bool renorm = inv != 0.0f;
#pragma unroll
                for (int jj = 0; jj < G::IT; ++jj) {
                    const int a = jj * 64 + lane;
                    const float r = a < hw ? srow[i * hw + a] : 0.0f;
                    S.policy[tn[i] * ROWP + a] = renorm ? r * inv : r;
                }
                if (lane == 0) S.hdr[tn[i]].has_policy = 1;
            }
        }
        __syncthreads(); // (the next batch overwrites the rows)
    }
}

// part 2, one wave per TREE: backups in simulation order (the f32 sums of w depend on the order), pme.rs:229,264.
// The requests of a tree and round are (almost always) children of ONE leaf, so their backups climb the SAME path: it is walked once
// (lane i keeps the (n, w) slot of the path's level i), every request's value is then added level-parallel in registers -- each
// level's w still receives its addends one request after the other, in simulation order, with the sign of its distance, exactly as
// Node::propagate applies them -- and stored once.  Per request only its own (fresh) slot in the leaf's table is touched, by its own
// lane.  That is 3 dependent loads per level of the path instead of per level AND request.  Requests with different parents (a
// backup of a terminal node between them moved the descent) take the plain per-request walk.
// (device function: k_scatter proper, and the head of k_round when the run loop defers a round's backups into the next round's kernel --
//  the backups then warm the very nodes the descent reads, and the round needs one launch less)
// The backups of a round's requests (node.rs:83-99, in request order).  A tree's requests are children of ONE leaf in ~95 % of its rounds; in the others the leaf filled up
// (or a terminal child sent the next simulation elsewhere) in mid-round and they fall into two or three SEGMENTS of consecutive requests with one parent each.  A segment's
// requests share the path above their parent, so its backups are taken together: lane i holds the path's level i, the values are added level by level in request order
// (the same additions in the same order as one backup after the other), one store per slot.  Segments follow each other in request order with the earlier one's stores landed
// first (their paths share the slots near the root).  Round 5: before, any round with a second parent took all K backups one by one, a walk of dependent loads and a store
// fence each -- 4-7 % of the trees in EVERY round from ply 0 on (a leaf fills up every 14 rounds), and with every tree's wave resident at once those were k_round's time.
template <int N>
__device__ inline void scatter_tree(const Store& S, const Tree<N>& T, const TreeState& ts, Regs& R, const float* __restrict__ V) {
    using G = Geo<N>;
    constexpr int ROWP = G::ROWP;
    const int lane = LANE, nreq = (int)ts.n_req;
    // every request's node, its parent and action (one lane per request; n_req <= KMAX = 64)
    int x = 0, par = -1 - lane, act = 0;
    float val = 0.0f;
    if (lane < nreq) {
        x = (int)T.req[lane];
        const NodeHdr hx = T.hdr[x];
        par = (int)hx.parent;
        act = (int)hx.action;
        val = V[(size_t)ts.req_base + lane];
    }
    const int prev_par = __shfl_up(par, 1, 64);
    unsigned long long rem = __ballot(lane < nreq && (lane == 0 || par != prev_par)); // segment starts
#if SCATTER_SEGMENTS == 0
    if (__popcll(rem) > 1) { // (A-B builds: the round-4 form -- a second parent sends every backup of the round through the walk)
        for (uint32_t r = 0; r < ts.n_req; ++r) {
            backup<N>(T, R, T.req[r], -V[(size_t)ts.req_base + r]);
            __syncthreads();
            R.bytes += 8ull * G::HW + 4;
        }
        return;
    }
#endif
    while (rem) {
        const int s0 = __ffsll((long long)rem) - 1;
        rem &= rem - 1ULL;
        const int s1 = rem ? __ffsll((long long)rem) - 1 : nreq; // the segment = requests [s0, s1)
        const int leaf = __shfl(par, s0, 64);
        // the path above the leaf: level 0 = the leaf's slot in its parent's table, ... up to the child of the root
        int depth = 0;
        size_t my_slot = 0;
        uint32_t my_n = 0;
        float my_w = 0.0f;
        bool fits = true;
        {
            int y = leaf;
            while (y != 0) {
                const NodeHdr h = T.hdr[y];
                const size_t slot = (size_t)T.hdr[h.parent].table * ROWP + h.action;
                const uint32_t n = T.cn[slot];
                const float w = T.cw[slot];
                if (lane == depth) { my_slot = slot; my_n = n; my_w = w; }
                y = h.parent;
                depth += 1;
                if (depth >= 64) { fits = y == 0; break; }
            }
        }
        if (fits) {
            const size_t tab_leaf = (size_t)T.hdr[leaf].table * ROWP;
            // own slots: n += 1, w += -v (the first step of propagate), one lane per request
            if (lane >= s0 && lane < s1) {
                const size_t slot = tab_leaf + act;
                const uint32_t n = T.cn[slot] + 1u;
                const float w = T.cw[slot] + (-val);
                T.cn[slot] = n;
                T.cw[slot] = w;
            }
            // levels and root: the value alternates its sign with the distance; level i (distance i + 1 from the request) gets +v for even i
            float root_w = R.root_w;
            const float sgn_root = (depth & 1) ? -1.0f : 1.0f;
            for (int r = s0; r < s1; ++r) {
                const float v = __shfl(val, r, 64);
                if (lane < depth) { my_n += 1u; my_w += (lane & 1) ? -v : v; }
                root_w += sgn_root * v; // (+-v exactly: a multiplication by +-1)
            }
            if (lane < depth) { T.cn[my_slot] = my_n; T.cw[my_slot] = my_w; }
            R.root_n += (uint32_t)(s1 - s0);
            R.root_w = root_w;
            R.bytes += (unsigned long long)(s1 - s0) * (16ull * (unsigned long long)(depth + 1) + 16ull + 8ull * G::HW + 4ull);
            if (rem) __syncthreads(); // the next segment's path shares slots with this one: its loads come after these stores
        } else {
            for (int r = s0; r < s1; ++r) {
                backup<N>(T, R, (int)T.req[r], -V[(size_t)ts.req_base + (uint32_t)r]);
                __syncthreads();
                R.bytes += 8ull * G::HW + 4;
            }
        }
    }
}

template <int N>
__global__ __launch_bounds__(64) void k_scatter(Store S, int side, const float* __restrict__ V) {
    const int g = blockIdx.x;
    if (!S.gs[g].alive) return;
    const int t = side * S.games + g;
    const Tree<N> T(S, t);
    const TreeState ts = *T.ts;
    if (ts.n_req == 0) return;
    Regs R{ts.n_nodes, ts.n_tables, ts.root_n, 0u, ts.error, ts.root_w, 0ull};
    scatter_tree<N>(S, T, ts, R, V);
    if (LANE == 0) {
        TreeState o = ts;
        o.root_n = R.root_n; o.root_w = R.root_w; o.n_req = 0;
        *T.ts = o;
        atomicAdd(S.d_bytes, R.bytes);
    }
}

// ---------------------------------------------------------------------------------------------
// k_reset: Agent::new (agent.rs:16-35) for both trees of every game
// ---------------------------------------------------------------------------------------------
template <int N>
__global__ __launch_bounds__(64) void k_reset(Store S, const float* __restrict__ root_policy) {
    using G = Geo<N>;
    const int t = blockIdx.x;
    const int lane = LANE;
    const Tree<N> T(S, t);
#pragma unroll
    for (int j = 0; j < G::IT; ++j) {
        const int a = j * 64 + lane;
        T.pol[a] = a < G::HW ? root_policy[a] : 0.0f;
    }
    if (lane < 2 * G::NW) T.board[lane] = 0ULL;
    if (lane == 0) {
        NodeHdr h;
        h.parent = NONE16; h.table = NONE16; h.legal = (uint16_t)G::HW; h.nch = 0;
        h.action = NONE8; h.status = ST_IN_PROGRESS; h.turn = 0; h.has_policy = 1; h.pad = 0;
        T.hdr[0] = h;
        TreeState s{1u, 0u, 0u, 0.0f, 0u, 0u, 0u, 0u};
        *T.ts = s;
        if (t < S.games) {
            GameState gs{};
            gs.alive = 1; gs.status = ST_IN_PROGRESS; gs.last_action = -1; gs.mirror_idx = -1; gs.gid = t;
            S.gs[t] = gs;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// slots mode: harvest finished games, restart their slots (omok_selfplay_run_slots)
// ---------------------------------------------------------------------------------------------
// one workgroup: slots whose game is over and not yet harvested, in slot order -> record offsets (appended at *out_count), per-game meta
template <int N>
__global__ __launch_bounds__(1024) void k_harvest_scan(Store S, uint8_t* __restrict__ mask, long long* __restrict__ slot_off, long long* __restrict__ out_count,
                                                       SlotMeta* __restrict__ meta) {
    __shared__ int s_part[1024];
    __shared__ long long s_base;
    const int tid = threadIdx.x, per = (S.games + 1023) / 1024;
    int mine = 0;
    for (int i = 0; i < per; ++i) {
        const int g = tid * per + i;
        if (g < S.games) {
            const GameState gs = S.gs[g];
            if (!gs.alive && !gs.harvested) mine += gs.rp_len < Geo<N>::HW ? gs.rp_len : Geo<N>::HW;
        }
    }
    s_part[tid] = mine;
    __syncthreads();
    if (tid == 0) {
        int acc = 0;
        for (int i = 0; i < 1024; ++i) { const int v = s_part[i]; s_part[i] = acc; acc += v; }
        s_base = *out_count;
        *out_count = s_base + acc;
    }
    __syncthreads();
    long long off = s_base + s_part[tid];
    for (int i = 0; i < per; ++i) {
        const int g = tid * per + i;
        if (g >= S.games) break;
        const GameState gs = S.gs[g];
        const bool take = !gs.alive && !gs.harvested;
        mask[g] = take ? 1 : 0;
        if (take) {
            const int len = gs.rp_len < Geo<N>::HW ? gs.rp_len : Geo<N>::HW;
            slot_off[g] = off;
            meta[gs.gid] = SlotMeta{off, len, (int32_t)gs.status};
            off += len;
            S.gs[g].harvested = 1;
        }
    }
}

// one workgroup: harvested slots in slot order take the next game indices while any are left (deterministic: a prefix sum, no atomics)
__global__ __launch_bounds__(1024) void k_refill_assign(Store S, int32_t* __restrict__ next_gid, int total_games, int32_t* __restrict__ new_gid) {
    __shared__ int s_part[1024];
    __shared__ int s_base;
    const int tid = threadIdx.x, per = (S.games + 1023) / 1024;
    int mine = 0;
    for (int i = 0; i < per; ++i) {
        const int g = tid * per + i;
        if (g < S.games && !S.gs[g].alive && S.gs[g].harvested) mine += 1;
    }
    s_part[tid] = mine;
    __syncthreads();
    if (tid == 0) {
        int acc = 0;
        for (int i = 0; i < 1024; ++i) { const int v = s_part[i]; s_part[i] = acc; acc += v; }
        s_base = *next_gid;
        const int left = total_games - s_base;
        *next_gid = s_base + (acc < left ? acc : (left > 0 ? left : 0));
    }
    __syncthreads();
    int id = s_base + s_part[tid];
    for (int i = 0; i < per; ++i) {
        const int g = tid * per + i;
        if (g >= S.games) break;
        int v = -1;
        if (!S.gs[g].alive && S.gs[g].harvested) { v = id < total_games ? id : -1; id += 1; }
        new_gid[g] = v;
    }
}

// Agent::new for both trees of the slots that take a new game
template <int N>
__global__ __launch_bounds__(64) void k_refill_reset(Store S, const float* __restrict__ root_policy, const int32_t* __restrict__ new_gid) {
    using G = Geo<N>;
    const int t = blockIdx.x, g = t % S.games;
    const int gid = new_gid[g];
    if (gid < 0) return;
    const int lane = LANE;
    const Tree<N> T(S, t);
#pragma unroll
    for (int j = 0; j < G::IT; ++j) {
        const int a = j * 64 + lane;
        T.pol[a] = a < G::HW ? root_policy[a] : 0.0f;
    }
    if (lane < 2 * G::NW) T.board[lane] = 0ULL;
    if (lane == 0) {
        NodeHdr h;
        h.parent = NONE16; h.table = NONE16; h.legal = (uint16_t)G::HW; h.nch = 0;
        h.action = NONE8; h.status = ST_IN_PROGRESS; h.turn = 0; h.has_policy = 1; h.pad = 0;
        T.hdr[0] = h;
        TreeState s{1u, 0u, 0u, 0.0f, 0u, 0u, 0u, 0u};
        *T.ts = s;
        if (t < S.games) {
            GameState gs{};
            gs.alive = 1; gs.status = ST_IN_PROGRESS; gs.last_action = -1; gs.mirror_idx = -1; gs.gid = gid;
            S.gs[t] = gs;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// k_sample: Agent::compute_policy + sample_action (agent.rs:43-137), transition record
// ---------------------------------------------------------------------------------------------
template <int N>
__global__ __launch_bounds__(64) void k_sample(Store S, int side, int ply, float temperature, int threshold,
                                               uint64_t seed, int64_t game_offset, int32_t* __restrict__ actions) {
    using G = Geo<N>;
    constexpr int ROWP = G::ROWP, NW = G::NW;
    __shared__ float s_row[ROWP];
    __shared__ int s_action;
    const int g = blockIdx.x;
    const int lane = LANE;
    if (!S.gs[g].alive) {
        if (lane == 0) { actions[g] = -1; S.gs[g].last_action = -1; }
        return;
    }
    const int t = side * S.games + g;
    const Tree<N> T(S, t);
    const NodeHdr h0 = T.hdr[0];
    if (h0.table == NONE16 || h0.nch == 0) {
        if (lane == 0) { actions[g] = -1; S.gs[g].last_action = -1; T.ts->error |= 4u; }
        return;
    }
    const size_t tb = (size_t)h0.table * ROWP;
    uint32_t nv[G::IT];
    uint32_t tot = 0;
#pragma unroll
    for (int j = 0; j < G::IT; ++j) {
        const int a = j * 64 + lane;
        nv[j] = T.corder[tb + a] != NONE8 ? T.cn[tb + a] : 0u;
        tot += nv[j];
    }
    tot = wave_sum_u32(tot); // visit counts are integers < 2^24: the f32 sum of agent.rs:58-63 is exact
    const float sum = (float)tot;
    if (sum < F32_EPS) {
        if (lane == 0) { actions[g] = -1; S.gs[g].last_action = -1; T.ts->error |= 4u; }
        return;
    }
    const float sum_inv = __fdiv_rn(1.0f, sum);
    float pi[G::IT];
#pragma unroll
    for (int j = 0; j < G::IT; ++j) pi[j] = (float)nv[j] * sum_inv;
    int action;
    const int plies = S.gs[g].rp_len; // turn_counts[index] (trainer.rs:139): moves sampled so far = transitions recorded
    if (plies < threshold) { // Boltzmann (agent.rs:106-133)
        const float tinv = __fdiv_rn(1.0f, temperature);
#pragma unroll
        for (int j = 0; j < G::IT; ++j) {
            const int a = j * 64 + lane;
            s_row[a] = (a < G::HW && !(pi[j] < F32_EPS)) ? det_expf(pi[j] * tinv) : 0.0f;
        }
        __syncthreads();
        const float hsum = seq_sum(s_row, G::HW);
        const float hinv = __fdiv_rn(1.0f, hsum);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < G::IT; ++j) s_row[j * 64 + lane] *= hinv;
        __syncthreads();
        if (lane == 0) {
            const float total = seq_sum(s_row, G::HW);
            const U4 o = philox(seed, 0u, (uint32_t)S.gs[g].plies, (uint32_t)((game_offset + S.gs[g].gid) * 2 + side), RNG_SAMPLE); // (the game's own ply and id)
            const float u = (float)(o.x >> 8) * 5.9604644775390625e-8f;
            const float target = u * total;
            float cum = 0.0f;
            int chosen = -1, last_nz = 0;
            for (int a = 0; a < G::HW; ++a) {
                const float x = s_row[a];
                if (!(x > 0.0f)) continue;
                last_nz = a;
                cum += x;
                if (chosen < 0 && cum > target) chosen = a;
            }
            s_action = chosen < 0 ? last_nz : chosen;
        }
        __syncthreads();
        action = s_action;
    } else { // Best: last max by total_cmp over all cells (agent.rs:98-105)
        unsigned long long best = 0ULL;
#pragma unroll
        for (int j = 0; j < G::IT; ++j) {
            const int a = j * 64 + lane;
            if (a < G::HW) {
                const unsigned long long v = ((unsigned long long)total_key_biased(pi[j]) << 32) | (unsigned long long)a;
                best = v > best ? v : best;
            }
        }
        best = wave_max_u64(best);
        action = (int)(best & 0xFFFFFFFFu);
    }
    // transition record (trainer.rs:150-173): env before the move, unheated pi; z is set by k_advance
    const size_t rec = (size_t)g * G::HW + (size_t)plies;
    if (plies < G::HW) {
        if (lane < 2 * NW) S.rp_board[rec * (2 * NW) + lane] = T.board[lane];
#pragma unroll
        for (int j = 0; j < G::IT; ++j) S.rp_pi[rec * ROWP + j * 64 + lane] = pi[j];
        if (lane == 0) { S.rp_turn[rec] = h0.turn; S.rp_z[rec] = 0.0f; }
    }
    if (lane == 0) { actions[g] = action; S.gs[g].last_action = action; S.gs[g].external = 0; }
}

// Agent::compute_policy (agent.rs:43-77) of the side-to-move agent of every game: pi [G][HW], has [G] (0 = None: finished
// game, no children or no visits)
template <int N>
__global__ __launch_bounds__(64) void k_policy(Store S, int side, float* __restrict__ pi, uint8_t* __restrict__ has) {
    using G = Geo<N>;
    constexpr int ROWP = G::ROWP;
    const int g = blockIdx.x;
    const int lane = LANE;
    const Tree<N> T(S, side * S.games + g);
    const NodeHdr h0 = T.hdr[0];
    uint32_t nv[G::IT];
    uint32_t tot = 0;
    const bool any = S.gs[g].alive && h0.table != NONE16 && h0.nch != 0;
#pragma unroll
    for (int j = 0; j < G::IT; ++j) {
        const int a = j * 64 + lane;
        nv[j] = (any && T.corder[(size_t)h0.table * ROWP + a] != NONE8) ? T.cn[(size_t)h0.table * ROWP + a] : 0u;
        tot += nv[j];
    }
    tot = wave_sum_u32(tot);
    const float sum = (float)tot;
    const bool some = any && !(sum < F32_EPS);
    const float sum_inv = some ? __fdiv_rn(1.0f, sum) : 0.0f;
#pragma unroll
    for (int j = 0; j < G::IT; ++j) {
        const int a = j * 64 + lane;
        if (a < G::HW) pi[(size_t)g * G::HW + a] = some ? (float)nv[j] * sum_inv : 0.0f;
    }
    if (lane == 0) has[g] = some ? 1 : 0;
}

// omok_play_actions: externally chosen moves (gui/src/agent.rs:49-66, benchmark/src/agent.rs:34-50 style callers) take the
// place of k_sample.  flags[0] counts illegal moves (out of range / occupied cell: Option::None of place_stone,
// environment/src/lib.rs:105-107), flags[1] live games without a move.  Nothing is recorded: the trainer pushes a Transition
// only for moves it sampled (trainer.rs:138-173).
template <int N>
__global__ __launch_bounds__(256) void k_set_actions(Store S, int side, const int32_t* __restrict__ actions, uint32_t* __restrict__ flags) {
    using G = Geo<N>;
    constexpr int NW = G::NW;
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= S.games) return;
    GameState gs = S.gs[g];
    gs.last_action = -1;
    gs.external = 1;
    if (gs.alive) {
        const int a = actions[g];
        if (a < 0) atomicAdd(&flags[1], 1u);
        else if (a >= G::HW) atomicAdd(&flags[0], 1u);
        else {
            const uint64_t* bb = S.board + (size_t)(side * S.games + g) * (size_t)S.stride_nodes * (2 * NW); // root = node 0
            const uint64_t occ = bb[a >> 6] | bb[NW + (a >> 6)];
            if ((occ >> (a & 63)) & 1ULL) atomicAdd(&flags[0], 1u);
            else gs.last_action = a;
        }
    }
    S.gs[g] = gs;
}
__global__ __launch_bounds__(256) void k_clear_actions(Store S) {
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= S.games) return;
    S.gs[g].last_action = -1;
    S.gs[g].external = 0;
    S.gs[g].mirror_idx = -1;
}

// ---------------------------------------------------------------------------------------------
// k_mirror_scan: NN requests of ensure_action_exists (agent.rs:153-158) for every live game
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_mirror_scan(Store S, int side) {
    __shared__ uint32_t s_part[1024];
    __shared__ uint32_t s_carry;
    const int tid = threadIdx.x;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (int base = 0; base < S.games; base += 1024) {
        const int g = base + tid;
        const uint32_t cnt = (g < S.games && S.gs[g].alive && S.gs[g].last_action >= 0) ? 1u : 0u;
        s_part[tid] = cnt;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            const uint32_t v = tid >= off ? s_part[tid - off] : 0u;
            __syncthreads();
            s_part[tid] += v;
            __syncthreads();
        }
        if (g < S.games) {
            if (cnt) {
                const uint32_t d = s_part[tid] - 1u + s_carry;
                const int topp = (1 - side) * S.games + g; // the opponent's tree still has the pre-move root
                S.req_ref[d] = (uint32_t)topp << 16;
                S.req_aux[d] = (uint32_t)S.gs[g].last_action | (1u << 16); // add the stone, Opponent mode
                S.gs[g].mirror_idx = (int32_t)d;
            } else {
                S.gs[g].mirror_idx = -1;
            }
        }
        __syncthreads();
        if (tid == 1023) s_carry += s_part[1023];
        __syncthreads();
    }
    if (tid == 0) S.d_count[0] = (int32_t)s_carry;
}

// ---------------------------------------------------------------------------------------------
// transition (mcts/src/lib.rs:47-78): keep the chosen child's subtree by stable compaction
// ---------------------------------------------------------------------------------------------
template <int N>
__device__ int transition(const Store& S, const Tree<N>& T, int action, uint8_t* s_alive, uint16_t* s_nmap,
                          uint16_t* s_tmap) {
    using G = Geo<N>;
    constexpr int ROWP = G::ROWP, NW = G::NW;
    const int lane = LANE;
    const TreeState ts = *T.ts;
    const NodeHdr h0 = T.hdr[0];
    if (h0.table == NONE16) return -1;
    const size_t tb0 = (size_t)h0.table * ROWP;
    if (T.corder[tb0 + action] == NONE8) return -1;
    const int c = T.cidx[tb0 + action];
    const float new_w = T.cw[tb0 + action];
    const NodeHdr hc = T.hdr[c];
    uint32_t new_n = 0; // lib.rs:65-71
    if (hc.table != NONE16) {
#pragma unroll
        for (int j = 0; j < G::IT; ++j) {
            const size_t s = (size_t)hc.table * ROWP + j * 64 + lane;
            new_n += T.corder[s] != NONE8 ? T.cn[s] : 0u;
        }
        new_n = wave_sum_u32(new_n);
    }
    const int nn = (int)ts.n_nodes, nt = (int)ts.n_tables;
    // 1. mark the subtree (parent index < child index, so one ascending pass resolves it)
    for (int i = lane; i < nn; i += 64) s_alive[i] = i == c ? 1 : (i < c ? 0 : 2);
    __syncthreads();
    for (int base = c + 1; base < nn; base += 64) {
        const int i = base + lane;
        const int par = i < nn ? (int)T.hdr[i].parent : 0;
        bool pending = i < nn;
        while (__any(pending)) {
            if (pending) {
                const uint8_t v = s_alive[par];
                if (v != 2) { s_alive[i] = v; pending = false; }
            }
            __syncthreads();
        }
    }
    // 2. stable renumbering
    int cnt = 0;
    for (int base = 0; base < nn; base += 64) {
        const int i = base + lane;
        const bool a = i < nn && s_alive[i] == 1;
        const unsigned long long m = __ballot(a);
        if (i < nn) s_nmap[i] = a ? (uint16_t)(cnt + __popcll(m & ((1ULL << lane) - 1ULL))) : NONE16;
        cnt += __popcll(m);
    }
    const int new_nodes = cnt;
    __syncthreads();
    cnt = 0;
    for (int base = 0; base < nt; base += 64) {
        const int k = base + lane;
        const bool a = k < nt && s_alive[T.owner[k]] == 1;
        const unsigned long long m = __ballot(a);
        if (k < nt) s_tmap[k] = a ? (uint16_t)(cnt + __popcll(m & ((1ULL << lane) - 1ULL))) : NONE16;
        cnt += __popcll(m);
    }
    const int new_tables = cnt;
    __syncthreads();
    // 3. move nodes, ascending (dst <= src, so an unread survivor is never overwritten)
    for (int base = 0; base < nn; base += 64) {
        const int i = base + lane;
        const bool a = i < nn && s_alive[i] == 1;
        NodeHdr h{};
        uint64_t bw[2 * NW];
        int dst = 0;
        if (a) {
            h = T.hdr[i];
            dst = s_nmap[i];
#pragma unroll
            for (int q = 0; q < 2 * NW; ++q) bw[q] = T.board[(size_t)i * (2 * NW) + q];
            h.parent = i == c ? NONE16 : s_nmap[h.parent];
            if (h.table != NONE16) h.table = s_tmap[h.table];
        }
        __syncthreads();
        if (a) {
            T.hdr[dst] = h;
#pragma unroll
            for (int q = 0; q < 2 * NW; ++q) T.board[(size_t)dst * (2 * NW) + q] = bw[q];
        }
        unsigned long long m = __ballot(a && h.has_policy && dst != i);
        while (m) { // policy rows: the whole wave copies a row; four rows' loads go out before their stores (one row at a time was a dependent round trip per surviving
                    // node, most of this kernel's time).  Ascending with dst <= src: a store never reaches a row that is still to be read (src' > src >= dst).
            constexpr int PB = 4;
            int srcs[PB], dsts[PB];
            float v[PB][G::IT];
#pragma unroll
            for (int q = 0; q < PB; ++q) {
                srcs[q] = -1;
                dsts[q] = 0;
                if (m) {
                    const int l = __ffsll((long long)m) - 1;
                    m &= m - 1;
                    srcs[q] = base + l;
                    dsts[q] = s_nmap[base + l];
                }
            }
#pragma unroll
            for (int q = 0; q < PB; ++q)
                if (srcs[q] >= 0) {
#pragma unroll
                    for (int j = 0; j < G::IT; ++j) v[q][j] = T.pol[(size_t)srcs[q] * ROWP + j * 64 + lane];
                }
#pragma unroll
            for (int q = 0; q < PB; ++q)
                if (srcs[q] >= 0) {
#pragma unroll
                    for (int j = 0; j < G::IT; ++j) T.pol[(size_t)dsts[q] * ROWP + j * 64 + lane] = v[q][j];
                }
        }
        __syncthreads();
    }
    // 4. move tables, ascending
    for (int k = 0; k < nt; ++k) {
        const int d = s_tmap[k];
        if (d == NONE16) continue;
        uint32_t vn[G::IT];
        float vw[G::IT];
        uint16_t vi[G::IT];
        uint8_t vo[G::IT];
#pragma unroll
        for (int j = 0; j < G::IT; ++j) {
            const size_t s = (size_t)k * ROWP + j * 64 + lane;
            vn[j] = T.cn[s]; vw[j] = T.cw[s]; vi[j] = T.cidx[s]; vo[j] = T.corder[s];
        }
        const uint16_t own = s_nmap[T.owner[k]];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < G::IT; ++j) {
            const size_t s = (size_t)d * ROWP + j * 64 + lane;
            T.cn[s] = vn[j]; T.cw[s] = vw[j]; T.corder[s] = vo[j];
            T.cidx[s] = vo[j] != NONE8 ? s_nmap[vi[j]] : (uint16_t)0;
        }
        if (lane == 0) T.owner[d] = own;
        __syncthreads();
    }
    if (lane == 0) {
        TreeState o = ts;
        o.n_nodes = (uint32_t)new_nodes; o.n_tables = (uint32_t)new_tables; o.root_n = new_n; o.root_w = new_w;
        *T.ts = o;
    }
    __syncthreads();
    return 0;
}

// Agent::ensure_action_exists (agent.rs:144-197) on tree T: p_row = evaluate_p of (root position + action, Opponent mode)
template <int N>
__device__ void ensure_action_exists(const Store& S, const Tree<N>& T, int action, const float* __restrict__ p_row, float* s_row) {
    using G = Geo<N>;
    constexpr int ROWP = G::ROWP, NW = G::NW;
    const int lane = LANE;
    const NodeHdr h0 = T.hdr[0];
    uint64_t bb[2 * NW];
#pragma unroll
    for (int i = 0; i < 2 * NW; ++i) bb[i] = T.board[i];
#pragma unroll
    for (int j = 0; j < G::IT; ++j) {
        const int a = j * 64 + lane;
        const bool empty = a < G::HW && !(((bb[j] | bb[NW + j]) >> lane) & 1ULL);
        s_row[a] = (empty && a != action) ? p_row[a] : 0.0f; // agent.rs:166-171
    }
    __syncthreads();
    const float sum = seq_sum(s_row, G::HW);
    const bool renorm = F32_EPS <= sum;
    const float inv = renorm ? __fdiv_rn(1.0f, sum) : 1.0f;
    const bool exists = h0.table != NONE16 && T.corder[(size_t)h0.table * ROWP + action] != NONE8;
    if (!exists) { // node.rs:69-71 returns None when the child is already there
        const TreeState ts = *T.ts;
        Regs R{ts.n_nodes, ts.n_tables, ts.root_n, 0u, ts.error, ts.root_w, 0ull};
        if (!get_bit<NW>(bb, action) && !get_bit<NW>(bb + NW, action)) (void)place_and_status<N>(bb, h0.turn, h0.legal, action);
        NodeHdr h0m = h0;
        const int idx = add_child<N>(S, T, R, 0, h0m, action, bb, ST_IN_PROGRESS, 1 - h0.turn, 1);
        if (idx >= 0) {
#pragma unroll
            for (int j = 0; j < G::IT; ++j) {
                const int a = j * 64 + lane;
                const float x = s_row[a];
                T.pol[(size_t)idx * ROWP + a] = renorm ? x * inv : x;
            }
        }
        if (lane == 0) {
            TreeState o = ts;
            o.n_nodes = R.n_nodes; o.n_tables = R.n_tables; o.error = R.error;
            *T.ts = o;
        }
    }
    __syncthreads();
}

template <int N>
__global__ __launch_bounds__(64) void k_advance(Store S, int side, const float* __restrict__ P) {
    using G = Geo<N>;
    constexpr int ROWP = G::ROWP, NW = G::NW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* s_row = (float*)smem;
    uint16_t* s_nmap = (uint16_t*)(smem + ROWP * 4);
    uint16_t* s_tmap = s_nmap + S.cap_nodes;
    uint8_t* s_alive = (uint8_t*)(s_tmap + S.cap_tables);
    const int g = blockIdx.x;
    const int lane = LANE;
    const GameState gs = S.gs[g];
    if (!gs.alive || gs.last_action < 0) return;
    const int action = gs.last_action;
    const Tree<N> own(S, side * S.games + g);
    const Tree<N> opp(S, (1 - side) * S.games + g);
    // ---- ensure_action_exists (agent.rs:144-197) on the pre-move root: always for the opposite agent (trainer.rs:163-166);
    //      an externally supplied move need not be in the mover's own tree either, so it gets the same treatment there (both
    //      agents hold the same position, hence the same evaluate_p row) ----
    if (gs.external) ensure_action_exists<N>(S, own, action, P + (size_t)gs.mirror_idx * ROWP, s_row);
    ensure_action_exists<N>(S, opp, action, P + (size_t)gs.mirror_idx * ROWP, s_row);
    // ---- agent.play_action (agent.rs:206-232): status from the real rules on the root board ----
    int status;
    {
        const NodeHdr h0 = own.hdr[0];
        uint64_t bb[2 * NW];
#pragma unroll
        for (int i = 0; i < 2 * NW; ++i) bb[i] = own.board[i];
        status = place_and_status<N>(bb, h0.turn, h0.legal, action);
    }
    int err = 0;
    if (transition<N>(S, own, action, s_alive, s_nmap, s_tmap) != 0) err |= 16;
    if (transition<N>(S, opp, action, s_alive, s_nmap, s_tmap) != 0) err |= 32;
    if (lane == 0) {
        if (err) own.ts->error |= (uint32_t)err;
        GameState o = gs;
        if (!gs.external) { // transitions.push (trainer.rs:169-173): only moves the trainer sampled
            if (gs.rp_len < G::HW)
                S.rp_z[(size_t)g * G::HW + gs.rp_len] = (status == ST_BLACK_WIN || status == ST_WHITE_WIN) ? 1.0f : 0.0f;
            o.rp_len = gs.rp_len + 1;
        }
        o.external = 0;
        o.plies = gs.plies + 1;
        o.status = (uint8_t)status;
        o.alive = status == ST_IN_PROGRESS ? 1 : 0;
        o.last_action = -1;
        o.mirror_idx = -1;
        S.gs[g] = o;
    }
}

// ---------------------------------------------------------------------------------------------
// request -> f32 NN input (encoder.rs:10-46), used by the step-wise test API
// ---------------------------------------------------------------------------------------------
template <int N>
__device__ inline void load_request_board(const Store& S, uint32_t ref, uint32_t aux, uint64_t* bb, int& turn, int& mode) {
    constexpr int NW = Geo<N>::NW;
    const int t = (int)(ref >> 16), node = (int)(ref & 0xFFFFu);
    const size_t tn = (size_t)t * (size_t)S.stride_nodes + (size_t)node;
    const NodeHdr h = S.hdr[tn];
#pragma unroll
    for (int i = 0; i < 2 * NW; ++i) bb[i] = S.board[tn * (2 * NW) + i];
    turn = h.turn;
    mode = 0;
    if (aux != 0xFFFFFFFFu) { // env.clone() + place_stone(action) (agent.rs:154-155)
        const int action = (int)(aux & 0xFFFFu);
        mode = (int)(aux >> 16) & 1;
        if (!get_bit<NW>(bb, action) && !get_bit<NW>(bb + NW, action)) {
#pragma unroll
            for (int i = 0; i < NW; ++i) {
                const uint64_t bit = (action >> 6) == i ? (1ULL << (action & 63)) : 0ULL;
                bb[i] |= turn == 0 ? bit : 0ULL;
                bb[NW + i] |= turn == 0 ? 0ULL : bit;
            }
            turn = 1 - turn;
        }
    }
}

template <int N>
__global__ __launch_bounds__(256) void k_encode_requests(Store S, float* __restrict__ out, int max_b) {
    using G = Geo<N>;
    constexpr int NW = G::NW;
    const int b = blockIdx.x;
    if (b >= S.d_count[0] || b >= max_b) return;
    uint64_t bb[2 * NW];
    int turn, mode;
    load_request_board<N>(S, S.req_ref[b], S.req_aux[b], bb, turn, mode);
    const int persp = mode == 0 ? turn : 1 - turn; // encoder.rs:24-27
    uint64_t mine[NW], theirs[NW];
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        mine[i] = persp == 0 ? bb[i] : bb[NW + i];
        theirs[i] = persp == 0 ? bb[NW + i] : bb[i];
    }
    for (int m = threadIdx.x; m < 3 * G::HW; m += blockDim.x) {
        float v;
        if (m < 2 * G::HW) {
            const int cell = m >> 1;
            v = (m & 1) ? (float)get_bit<NW>(theirs, cell) : (float)get_bit<NW>(mine, cell);
        } else {
            v = turn == 0 ? 1.0f : 0.0f; // encoder.rs:34-37
        }
        out[(size_t)b * 3 * G::HW + m] = v;
    }
}

// ---------------------------------------------------------------------------------------------
// environment crate on device, batched (environment/src/lib.rs:73-166; encoder.rs:10-46)
// ---------------------------------------------------------------------------------------------
template <int N>
__global__ __launch_bounds__(64) void k_env_play(const int32_t* __restrict__ moves, int len, int32_t* __restrict__ status_out,
                                                 uint8_t* __restrict__ boards, uint8_t* __restrict__ turns, uint16_t* __restrict__ legal) {
    using G = Geo<N>;
    constexpr int NW = G::NW;
    const int b = blockIdx.x;
    const int lane = LANE;
    uint64_t bb[2 * NW];
#pragma unroll
    for (int i = 0; i < 2 * NW; ++i) bb[i] = 0ULL;
    int turn = 0, lg = G::HW;
    for (int i = 0; i < len; ++i) {
        const int a = moves[(size_t)b * len + i];
        int st = -1;
        if (a >= 0 && a < G::HW && !get_bit<NW>(bb, a) && !get_bit<NW>(bb + NW, a)) {
            st = place_and_status<N>(bb, turn, lg, a);
            lg -= 1;
            turn = 1 - turn;
        }
        if (lane == 0) status_out[(size_t)b * len + i] = st;
    }
#pragma unroll
    for (int j = 0; j < G::IT; ++j) {
        const int a = j * 64 + lane;
        if (a < G::HW) boards[(size_t)b * G::HW + a] = (uint8_t)(((bb[j] >> lane) & 1ULL) ? 1 : (((bb[NW + j] >> lane) & 1ULL) ? 2 : 0));
    }
    if (lane == 0) { turns[b] = (uint8_t)turn; legal[b] = (uint16_t)lg; }
}

// Environment::place_stone (environment/src/lib.rs:104-166) on caller-held environments: boards [B][HW] Stone bytes, turns [B],
// legal [B] are updated in place; status [B] = Option<GameStatus> (-1 = None: out of range or occupied, nothing changes)
template <int N>
__global__ __launch_bounds__(64) void k_env_place(uint8_t* __restrict__ boards, uint8_t* __restrict__ turns, uint16_t* __restrict__ legal,
                                                  const int32_t* __restrict__ actions, int32_t* __restrict__ status_out) {
    using G = Geo<N>;
    constexpr int NW = G::NW;
    const int b = blockIdx.x;
    const int lane = LANE;
    uint64_t bb[2 * NW];
#pragma unroll
    for (int j = 0; j < G::IT; ++j) {
        const int a = j * 64 + lane;
        const int s = a < G::HW ? boards[(size_t)b * G::HW + a] : 0;
        bb[j] = __ballot(s == 1);
        bb[NW + j] = __ballot(s == 2);
    }
    const int turn = turns[b] & 1, lg = legal[b];
    const int a = actions[b];
    int st = -1;
    if (a >= 0 && a < G::HW && !get_bit<NW>(bb, a) && !get_bit<NW>(bb + NW, a)) {
        st = place_and_status<N>(bb, turn, lg, a);
        if (lane == 0) {
            boards[(size_t)b * G::HW + a] = (uint8_t)(turn == 0 ? 1 : 2);
            turns[b] = (uint8_t)(1 - turn);
            legal[b] = (uint16_t)(lg - 1);
        }
    }
    if (lane == 0) status_out[b] = st;
}

template <int N>
__global__ __launch_bounds__(256) void k_encode_boards(const uint8_t* __restrict__ boards, const uint8_t* __restrict__ turns,
                                                       int mode, float* __restrict__ out) {
    using G = Geo<N>;
    const int b = blockIdx.x;
    const int turn = turns[b];
    const int persp = mode == 0 ? turn : 1 - turn;
    const int mine = persp == 0 ? 1 : 2;
    for (int m = threadIdx.x; m < 3 * G::HW; m += blockDim.x) {
        float v;
        if (m < 2 * G::HW) {
            const int s = boards[(size_t)b * G::HW + (m >> 1)];
            v = s == 0 ? 0.0f : (((m & 1) == 0) == (s == mine) ? 1.0f : 0.0f);
        } else {
            v = turn == 0 ? 1.0f : 0.0f;
        }
        out[(size_t)b * 3 * G::HW + m] = v;
    }
}

// replay tuples packed for an RCCL gather: board u8[HW], turn u8, pad (zero) to 4, pi f32[HW], z f32; games in id order
// (offsets = exclusive scan of the transition counts), transitions in play order: the packed buffer is deterministic
template <int N>
__global__ __launch_bounds__(64) void k_replay_pack(Store S, const long long* __restrict__ offsets, uint8_t* __restrict__ dst, long long cap,
                                                    const uint8_t* __restrict__ mask) {
    using G = Geo<N>;
    constexpr int NW = G::NW, ROWP = G::ROWP;
    constexpr int BRD = (G::HW + 1 + 3) / 4 * 4, REC = BRD + 4 * G::HW + 4;
    const int g = blockIdx.x;
    const int lane = LANE;
    if (mask && !mask[g]) return; // (harvest: only the slots k_harvest_scan picked)
    const int plies = S.gs[g].rp_len < G::HW ? S.gs[g].rp_len : G::HW;
    const long long base = offsets[g];
    for (int p = 0; p < plies; ++p) {
        if (base + p >= cap) break;
        uint8_t* r = dst + (size_t)(base + p) * REC;
        const size_t rec = (size_t)g * G::HW + p;
#pragma unroll
        for (int j = 0; j < G::IT; ++j) {
            const int a = j * 64 + lane;
            if (a < G::HW) {
                const uint64_t bw = S.rp_board[rec * (2 * NW) + j], ww = S.rp_board[rec * (2 * NW) + NW + j];
                r[a] = (uint8_t)(((bw >> lane) & 1ULL) ? 1 : (((ww >> lane) & 1ULL) ? 2 : 0));
                ((float*)(r + BRD))[a] = S.rp_pi[rec * ROWP + a];
            }
        }
        if (lane < BRD - G::HW) r[G::HW + lane] = lane == 0 ? S.rp_turn[rec] : (uint8_t)0;
        if (lane == 0) ((float*)(r + BRD))[G::HW] = S.rp_z[rec];
    }
}

// Replay post-processing of Trainer::train (src/trainer.rs:207-324) on the device, one wave per transition:
//   z back-fill: walking the game backwards from its last transition, z alternates sign (:209-214);
//   five augmented copies per transition, in the reference's order rotate_90, rotate_180, rotate_270, flip_horizontal,
//   flip_vertical of board and policy (:222-318, src/utils.rs:1-64), same z.
// Output per game (game-id order; the reference appends games in completion order): the L back-filled transitions
// (replay_memory.extend(transitions), :320), then the 5L augmented ones, transition-major (:321).  Record layout as in
// k_replay_pack.  offsets[g] = first record of game g (exclusive scan of 6 * plies).
template <int N>
__global__ void k_replay_offsets(Store S, int per_transition, long long* __restrict__ offsets) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    long long acc = 0;
    for (int g = 0; g < S.games; ++g) {
        offsets[g] = acc;
        const int plies = S.gs[g].rp_len < Geo<N>::HW ? S.gs[g].rp_len : Geo<N>::HW;
        acc += (long long)per_transition * plies;
    }
    offsets[S.games] = acc;
}

template <int N>
__device__ inline int aug_src(int k, int a) { // source cell of destination cell a under transform k (src/utils.rs)
    const int i = a / N, j = a % N;
    switch (k) {
        case 1: return (N - j - 1) * N + i;           // rotate_90   :7-11
        case 2: return (N - i - 1) * N + (N - j - 1); // rotate_180  :20-24
        case 3: return j * N + (N - i - 1);           // rotate_270  :33-37
        case 4: return i * N + (N - j - 1);           // flip_horizontal :46-50
        case 5: return (N - i - 1) * N + j;           // flip_vertical   :59-63
        default: return a;
    }
}

template <int N>
__global__ __launch_bounds__(64) void k_replay_augment(Store S, const long long* __restrict__ offsets, int game_first, long long base_sub,
                                                       uint8_t* __restrict__ dst, long long cap) {
    using G = Geo<N>;
    constexpr int NW = G::NW, ROWP = G::ROWP;
    constexpr int BRD = (G::HW + 1 + 3) / 4 * 4, REC = BRD + 4 * G::HW + 4;
    const int g = game_first + blockIdx.y; // one wave per transition: grid = (HW plies, games)
    const int lane = LANE;
    const int L = S.gs[g].rp_len < G::HW ? S.gs[g].rp_len : G::HW;
    const int p = blockIdx.x;
    if (p >= L) return;
    const long long base = offsets[g] - base_sub;
    const float z_last = S.rp_z[(size_t)g * G::HW + (L - 1)];
    {
        const size_t rec = (size_t)g * G::HW + p;
        const float z = ((L - 1 - p) & 1) ? -z_last : z_last; // transition.z = z; z = -z; (:211-214), incl. the sign of zero
        uint64_t bw[NW], ww[NW];
#pragma unroll
        for (int i = 0; i < NW; ++i) { bw[i] = S.rp_board[rec * (2 * NW) + i]; ww[i] = S.rp_board[rec * (2 * NW) + NW + i]; }
        const uint8_t turn = S.rp_turn[rec];
        for (int k = 0; k < 6; ++k) {
            const long long idx = k == 0 ? base + p : base + L + 5LL * p + (k - 1);
            if (idx >= cap) continue;
            uint8_t* r = dst + (size_t)idx * REC;
#pragma unroll
            for (int j = 0; j < G::IT; ++j) {
                const int a = j * 64 + lane;
                if (a < G::HW) {
                    const int s = aug_src<N>(k, a);
                    uint64_t b = bw[0], w = ww[0];
#pragma unroll
                    for (int i = 1; i < NW; ++i) { b = (s >> 6) == i ? bw[i] : b; w = (s >> 6) == i ? ww[i] : w; }
                    r[a] = (uint8_t)(((b >> (s & 63)) & 1ULL) ? 1 : (((w >> (s & 63)) & 1ULL) ? 2 : 0));
                    ((float*)(r + BRD))[a] = S.rp_pi[rec * ROWP + s];
                }
            }
            if (lane < BRD - G::HW) r[G::HW + lane] = lane == 0 ? turn : (uint8_t)0; // env.turn is cloned unchanged (:224); pad bytes zero
            if (lane == 0) ((float*)(r + BRD))[G::HW] = z;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
#define DISPATCH_N(n, CALL9, CALL15) \
    do { if ((n) == 9) { CALL9; } else { CALL15; } } while (0)

size_t advance_lds_bytes(int cap_nodes, int cap_tables) {
    return 256 * 4 + (size_t)cap_nodes * 2 + (size_t)cap_tables * 2 + (size_t)cap_nodes + 16;
}

void launch_reset(int n, const Store& S, const float* rp, hipStream_t st) {
    DISPATCH_N(n, (k_reset<9><<<2 * S.games, 64, 0, st>>>(S, rp)), (k_reset<15><<<2 * S.games, 64, 0, st>>>(S, rp)));
}
void launch_round(int n, const Store& S, const RoundArgs& a, hipStream_t st) {
    DISPATCH_N(n, (k_round<9><<<S.games, 64, 0, st>>>(S, a)), (k_round<15><<<S.games, 64, 0, st>>>(S, a)));
}
void launch_round_shared(int n, const Store& S, const RoundArgs& a, int rounds_total, int group, int waves, uint16_t* sh_req, uint32_t* sh_cnt,
                         hipStream_t st, uint8_t* rec_order, uint32_t* rec_pos) {
    DISPATCH_N(n, (k_round_shared<9><<<1, waves * 64, 0, st>>>(S, a, rounds_total, group, sh_req, sh_cnt, rec_order, rec_pos)),
               (k_round_shared<15><<<1, waves * 64, 0, st>>>(S, a, rounds_total, group, sh_req, sh_cnt, rec_order, rec_pos)));
    k_scan_shared<<<1, 64, 0, st>>>(S, a.side, waves, sh_req, sh_cnt);
}
void launch_scatter_shared(int n, const Store& S, int side, const float* p, const float* v, int max_count, int waves, const uint16_t* sh_req,
                           const uint32_t* sh_cnt, hipStream_t st, uint8_t* rec_order, uint32_t* rec_pos) {
    const int grid = max_count > 0 ? max_count : 1;
    DISPATCH_N(n, (k_scatter_policy<9><<<grid, 64, 0, st>>>(S, p, max_count)), (k_scatter_policy<15><<<grid, 64, 0, st>>>(S, p, max_count)));
    DISPATCH_N(n, (k_scatter_shared<9><<<1, waves * 64, 0, st>>>(S, side, v, sh_req, sh_cnt, rec_order, rec_pos)),
               (k_scatter_shared<15><<<1, waves * 64, 0, st>>>(S, side, v, sh_req, sh_cnt, rec_order, rec_pos)));
}
void launch_fill(const Store& S, int side, int K, hipStream_t st) { k_fill<<<(S.games * K + 255) / 256, 256, 0, st>>>(S, side, K); }
void launch_scan(int n, const Store& S, int side, int K, hipStream_t st, unsigned long long* evals, int32_t* zero_ptr, int zero_n, bool fill) {
    k_scan<<<1, 1024, 0, st>>>(S, side, evals, zero_ptr, zero_ptr ? zero_n : 0);
    if (fill) launch_fill(S, side, K, st);
}
void launch_scatter(int n, const Store& S, int side, const float* p, const float* v, int max_count, hipStream_t st, bool backups) {
    const int grid = max_count < 8192 ? (max_count > 0 ? max_count : 1) : 8192;
    DISPATCH_N(n, (k_scatter_policy<9><<<grid, 64, 0, st>>>(S, p, max_count)), (k_scatter_policy<15><<<grid, 64, 0, st>>>(S, p, max_count)));
    if (backups) launch_backups(n, S, side, v, st);
}
void launch_backups(int n, const Store& S, int side, const float* v, hipStream_t st) {
    DISPATCH_N(n, (k_scatter<9><<<S.games, 64, 0, st>>>(S, side, v)), (k_scatter<15><<<S.games, 64, 0, st>>>(S, side, v)));
}
void launch_softmax_scatter(int n, const Store& S, int side, const float* logits, int lrow, float* v, float* vpre, int max_count, hipStream_t st,
                            bool backups) {
    const int nb = (max_count + SSP_BATCH - 1) / SSP_BATCH; // (a wave takes SSP_BATCH requests at a time)
    const int grid = nb < 8192 ? (nb > 0 ? nb : 1) : 8192;
    DISPATCH_N(n, (k_softmax_scatter_policy<9><<<grid, 64, 0, st>>>(S, logits, lrow, v, vpre, max_count)),
               (k_softmax_scatter_policy<15><<<grid, 64, 0, st>>>(S, logits, lrow, v, vpre, max_count)));
    if (backups) launch_backups(n, S, side, v, st);
}
void launch_sample(int n, const Store& S, int side, int ply, float temperature, int threshold, uint64_t seed,
                   int64_t game_offset, int32_t* actions, hipStream_t st) {
    DISPATCH_N(n, (k_sample<9><<<S.games, 64, 0, st>>>(S, side, ply, temperature, threshold, seed, game_offset, actions)),
               (k_sample<15><<<S.games, 64, 0, st>>>(S, side, ply, temperature, threshold, seed, game_offset, actions)));
}
void launch_mirror_scan(int n, const Store& S, int side, hipStream_t st) { k_mirror_scan<<<1, 1024, 0, st>>>(S, side); }
void launch_advance(int n, const Store& S, int side, const float* p, hipStream_t st) {
    const size_t lds = advance_lds_bytes(S.cap_nodes, S.cap_tables);
    DISPATCH_N(n, (k_advance<9><<<S.games, 64, lds, st>>>(S, side, p)), (k_advance<15><<<S.games, 64, lds, st>>>(S, side, p)));
}
void launch_encode_requests(int n, const Store& S, float* out, int max_b, hipStream_t st) {
    if (max_b <= 0) return;
    DISPATCH_N(n, (k_encode_requests<9><<<max_b, 256, 0, st>>>(S, out, max_b)), (k_encode_requests<15><<<max_b, 256, 0, st>>>(S, out, max_b)));
}
void launch_env_play(int n, const int32_t* moves, int batch, int len, int32_t* status, uint8_t* boards, uint8_t* turns,
                     uint16_t* legal, hipStream_t st) {
    DISPATCH_N(n, (k_env_play<9><<<batch, 64, 0, st>>>(moves, len, status, boards, turns, legal)),
               (k_env_play<15><<<batch, 64, 0, st>>>(moves, len, status, boards, turns, legal)));
}
void launch_encode_boards(int n, const uint8_t* boards, const uint8_t* turns, int batch, int mode, float* out, hipStream_t st) {
    DISPATCH_N(n, (k_encode_boards<9><<<batch, 256, 0, st>>>(boards, turns, mode, out)),
               (k_encode_boards<15><<<batch, 256, 0, st>>>(boards, turns, mode, out)));
}
void launch_replay_offsets(int n, const Store& S, int per_transition, long long* offsets, hipStream_t st) {
    DISPATCH_N(n, (k_replay_offsets<9><<<1, 64, 0, st>>>(S, per_transition, offsets)), (k_replay_offsets<15><<<1, 64, 0, st>>>(S, per_transition, offsets)));
}
void launch_set_actions(int n, const Store& S, int side, const int32_t* actions, uint32_t* flags, hipStream_t st) {
    const int grid = (S.games + 255) / 256;
    DISPATCH_N(n, (k_set_actions<9><<<grid, 256, 0, st>>>(S, side, actions, flags)), (k_set_actions<15><<<grid, 256, 0, st>>>(S, side, actions, flags)));
}
void launch_clear_actions(const Store& S, hipStream_t st) { k_clear_actions<<<(S.games + 255) / 256, 256, 0, st>>>(S); }
void launch_policy(int n, const Store& S, int side, float* pi, uint8_t* has, hipStream_t st) {
    DISPATCH_N(n, (k_policy<9><<<S.games, 64, 0, st>>>(S, side, pi, has)), (k_policy<15><<<S.games, 64, 0, st>>>(S, side, pi, has)));
}
void launch_env_place(int n, uint8_t* boards, uint8_t* turns, uint16_t* legal, const int32_t* actions, int batch, int32_t* status, hipStream_t st) {
    DISPATCH_N(n, (k_env_place<9><<<batch, 64, 0, st>>>(boards, turns, legal, actions, status)),
               (k_env_place<15><<<batch, 64, 0, st>>>(boards, turns, legal, actions, status)));
}
// k_advance keeps 3 B per node and 2 B per table of re-rooting scratch in LDS: above 64 KiB the kernel needs the opt-in
// attribute; returns 0 when `bytes` fits the device limit and the attribute is set for both board sizes
int set_advance_lds_attribute(int n, size_t bytes) {
    const void* f = n == 9 ? (const void*)k_advance<9> : (const void*)k_advance<15>;
    return hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess ? 0 : -1;
}
void launch_replay_augment(int n, const Store& S, const long long* offsets, int game_first, int game_count, long long base_sub,
                           uint8_t* dst, long long cap, hipStream_t st) {
    if (game_count <= 0) return;
    DISPATCH_N(n, (k_replay_augment<9><<<dim3(81, game_count), 64, 0, st>>>(S, offsets, game_first, base_sub, dst, cap)),
               (k_replay_augment<15><<<dim3(225, game_count), 64, 0, st>>>(S, offsets, game_first, base_sub, dst, cap)));
}
void launch_replay_pack(int n, const Store& S, const long long* offsets, uint8_t* dst, long long cap, hipStream_t st) {
    DISPATCH_N(n, (k_replay_pack<9><<<S.games, 64, 0, st>>>(S, offsets, dst, cap, nullptr)),
               (k_replay_pack<15><<<S.games, 64, 0, st>>>(S, offsets, dst, cap, nullptr)));
}
void launch_harvest(int n, const Store& S, uint8_t* mask, long long* slot_off, long long* out_count, SlotMeta* meta, uint8_t* dst, long long cap,
                    hipStream_t st) {
    DISPATCH_N(n, (k_harvest_scan<9><<<1, 1024, 0, st>>>(S, mask, slot_off, out_count, meta)), (k_harvest_scan<15><<<1, 1024, 0, st>>>(S, mask, slot_off, out_count, meta)));
    DISPATCH_N(n, (k_replay_pack<9><<<S.games, 64, 0, st>>>(S, slot_off, dst, cap, mask)), (k_replay_pack<15><<<S.games, 64, 0, st>>>(S, slot_off, dst, cap, mask)));
}
void launch_refill(int n, const Store& S, const float* rp, int32_t* next_gid, int total_games, int32_t* new_gid, hipStream_t st) {
    k_refill_assign<<<1, 1024, 0, st>>>(S, next_gid, total_games, new_gid);
    DISPATCH_N(n, (k_refill_reset<9><<<2 * S.games, 64, 0, st>>>(S, rp, new_gid)), (k_refill_reset<15><<<2 * S.games, 64, 0, st>>>(S, rp, new_gid)));
}

} // namespace omok
