// engine.cpp — host side of libomok_mi355x.so: the opaque handle, HBM allocation, kernel
// orchestration and the C ABI of include/omok_mi355x.h.
//
// Mirrors the reference's host control flow:
//   omok_execute       <- ParallelMCTSExecutor::execute (alpha-zero/src/parallel_mcts_executor.rs:26-270)
//   omok_selfplay_run  <- Trainer::train self-play phase (src/trainer.rs:95-205)
// but enqueues a whole execute() (all rounds) on one HIP stream without host round trips: the
// live batch size stays on the device (Store::d_count) and the net kernels bound themselves by it.
#include "../../include/omok_mi355x.h"
#include "common.h"
#include "net.h"

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

using namespace omok;

// ---------------------------------------------------------------------------------------------
hipEvent_t Prof::get() {
    if (n_pool > 0) return pool[--n_pool];
    hipEvent_t e;
    hipEventCreate(&e);
    return e;
}
void Prof::begin(int cat, hipStream_t st) {
    if (!enabled || !active) return;
    if (n_items == cap_items) {
        if (cap_items >= 8192) resolve();
        else {
            cap_items = cap_items ? cap_items * 2 : 1024;
            items = (Item*)realloc(items, sizeof(Item) * cap_items);
        }
    }
    Item it{cat, in_round, get(), get()};
    hipEventRecord(it.a, st);
    items[n_items++] = it;
}
void Prof::end(hipStream_t st) {
    if (!enabled || !active || n_items == 0) return;
    hipEventRecord(items[n_items - 1].b, st);
}
void Prof::resolve() {
    if (n_items == 0) return;
    hipEventSynchronize(items[n_items - 1].b);
    if (cap_pool < n_pool + 2 * n_items) {
        cap_pool = n_pool + 2 * n_items + 64;
        pool = (hipEvent_t*)realloc(pool, sizeof(hipEvent_t) * cap_pool);
    }
    for (int i = 0; i < n_items; ++i) {
        float t = 0.f;
        if (hipEventElapsedTime(&t, items[i].a, items[i].b) == hipSuccess) {
            if (items[i].sampled) { ms_s[items[i].cat] += t; launches_s[items[i].cat] += 1; }
            else { ms[items[i].cat] += t; launches[items[i].cat] += 1; }
        }
        pool[n_pool++] = items[i].a;
        pool[n_pool++] = items[i].b;
    }
    n_items = 0;
}
void Prof::destroy() {
    resolve();
    for (int i = 0; i < n_pool; ++i) hipEventDestroy(pool[i]);
    free(pool);
    free(items);
    pool = nullptr; items = nullptr; n_pool = cap_pool = n_items = cap_items = 0;
}

// ---------------------------------------------------------------------------------------------
struct omok_engine {
    omok_config cfg{};
    int n = 0, hw = 0, rowp = 0, nw = 0, T = 0;
    Store S{};
    Net net{};
    Prof prof{};
    hipStream_t st = nullptr;
    std::vector<void*> allocs;
    std::string err;
    int ply = 0;
    bool reset_done = false;
    bool sampled = false;
    uint64_t episode = 0;  // index of the RNG stream the NEXT omok_selfplay_reset takes (one reset = one trainer iteration)
    uint64_t key = 0;      // Philox key of the current episode: cfg.seed + episode * 0x9E3779B97F4A7C15
    int round_reqs = -1;   // step-wise API state
    int round_cap = 0;     // alive * batch_size of the generated round: the net's max_count (same as omok_execute uses)
    int mirror_reqs = -1;
    float* d_root_policy = nullptr;
    int32_t* d_actions = nullptr;
    uint32_t* d_error = nullptr; // [0] error bits, [1] alive count
    unsigned long long* d_evals = nullptr;
    uint16_t* d_sh_req = nullptr;       // [MAX_TREE_WAVES][KMAX] requests of the waves of a shared-tree round group
    uint32_t* d_sh_cnt = nullptr;       // [2 * KMAX] their counts | first-request offsets
    uint32_t* d_flags = nullptr;        // [0] illegal external moves, [1] live games without a move (omok_play_actions)
    float* d_pi = nullptr;              // [G][HW] omok_compute_policy
    uint8_t* d_has = nullptr;           // [G]
    long long* d_aug_offsets = nullptr; // [games + 1] first augmented-replay record of every game
    uint8_t* d_aug_scratch = nullptr;   // one game's 6 * HW records (omok_replay_augmented_game)
    // host-side stats
    double sims = 0, evals = 0, ply_games = 0, finished = 0;
    uint32_t peak_nodes = 0, peak_tables = 0;
    size_t bytes = 0;
};

static std::string g_create_error;

static int fail(omok_engine* e, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (e) e->err = buf; else g_create_error = buf;
    return code;
}

#define HIPCHK(e, call)                                                                        \
    do {                                                                                       \
        hipError_t _r = (call);                                                                \
        if (_r != hipSuccess) return fail(e, OMOK_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(_r)); \
    } while (0)

// every entry point runs on the engine's own device, whatever the calling thread's current device is
#define ENTER(e)                                    \
    do {                                            \
        if (!(e)) return OMOK_ERR_INVALID;          \
        HIPCHK(e, hipSetDevice((e)->cfg.device));   \
    } while (0)

template <typename Tp>
static int dalloc(omok_engine* e, Tp** p, size_t count) {
    void* q = nullptr;
    const size_t bytes = count * sizeof(Tp);
    hipError_t r = hipMalloc(&q, bytes ? bytes : 16);
    if (r != hipSuccess) return fail(e, OMOK_ERR_HIP, "hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(r));
    e->allocs.push_back(q);
    e->bytes += bytes;
    *p = (Tp*)q;
    return 0;
}

// tiny helper kernels --------------------------------------------------------------------------
__global__ void k_count_alive(Store S, uint32_t* d_error, unsigned long long* d_evals) {
    __shared__ uint32_t s_cnt, s_err, s_mn, s_mt;
    if (threadIdx.x == 0) { s_cnt = 0; s_err = 0; s_mn = 0; s_mt = 0; }
    __syncthreads();
    uint32_t c = 0, er = 0, mn = 0, mt = 0;
    for (int g = threadIdx.x; g < S.games; g += blockDim.x) {
        c += S.gs[g].alive ? 1u : 0u;
        const TreeState a = S.ts[g], b = S.ts[S.games + g];
        er |= a.error | b.error;
        mn = max(mn, max(a.n_nodes, b.n_nodes));
        mt = max(mt, max(a.n_tables, b.n_tables));
    }
    atomicAdd(&s_cnt, c);
    atomicOr(&s_err, er);
    atomicMax(&s_mn, mn);
    atomicMax(&s_mt, mt);
    __syncthreads();
    if (threadIdx.x == 0) { d_error[0] = s_err; d_error[1] = s_cnt; d_error[2] = s_mn; d_error[3] = s_mt; }
}
__global__ void k_add_evals(const int32_t* d_count, unsigned long long* d_evals) {
    if (threadIdx.x == 0 && blockIdx.x == 0) d_evals[0] += (unsigned long long)d_count[0];
}
__global__ void k_copy_root_policy(const float* p, float* dst, int hw, int rowp) {
    const int a = threadIdx.x;
    if (a < rowp) dst[a] = a < hw ? p[a] : 0.0f;
}
__global__ void k_pack_rows(const float* src, float* dst, int hw, int rowp, int rows) { // [rows][ROWP] -> [rows][HW]
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (size_t)rows * hw) dst[i] = src[(i / hw) * rowp + (i % hw)];
}
__global__ void k_unpack_rows(const float* src, float* dst, int hw, int rowp, int rows) { // [rows][HW] -> [rows][ROWP]
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (size_t)rows * rowp) dst[i] = (int)(i % rowp) < hw ? src[(i / rowp) * hw + (i % rowp)] : 0.0f;
}

// ---------------------------------------------------------------------------------------------
extern "C" const char* omok_last_error(const omok_engine* e) { return e ? e->err.c_str() : g_create_error.c_str(); }

extern "C" void omok_destroy(omok_engine* e) {
    if (!e) return;
    hipSetDevice(e->cfg.device);
    if (e->st) hipStreamSynchronize(e->st);
    e->prof.destroy();
    net_free(e->net);
    for (void* p : e->allocs) hipFree(p);
    if (e->st) hipStreamDestroy(e->st);
    delete e;
}

extern "C" int omok_create(const omok_config* cfg, omok_engine** out) {
    if (!cfg || !out) return fail(nullptr, OMOK_ERR_INVALID, "null argument");
    *out = nullptr;
    if (cfg->board_size != 9 && cfg->board_size != 15)
        return fail(nullptr, OMOK_ERR_INVALID, "board_size must be 9 or 15 (got %d)", cfg->board_size);
    if (cfg->games < 1 || cfg->games > MAX_GAMES) return fail(nullptr, OMOK_ERR_INVALID, "games must be in [1, %d]", MAX_GAMES);
    if (cfg->max_nodes < 2 || cfg->max_nodes > OMOK_MAX_ARENA || cfg->max_tables < 1 || cfg->max_tables > OMOK_MAX_ARENA)
        return fail(nullptr, OMOK_ERR_INVALID, "max_nodes must be in [2, %d] and max_tables in [1, %d]", OMOK_MAX_ARENA, OMOK_MAX_ARENA);
    if (cfg->max_batch_k < 1 || cfg->max_batch_k > KMAX) return fail(nullptr, OMOK_ERR_INVALID, "max_batch_k must be in [1, 64]");
    if (cfg->net_mode != OMOK_NET_F16X3 && cfg->net_mode != OMOK_NET_F32 && cfg->net_mode != OMOK_NET_F16X3_ROWS && cfg->net_mode != OMOK_NET_F16X3_FP6 &&
        cfg->net_mode != OMOK_NET_F16X3_F16 && cfg->net_mode != OMOK_NET_F16X3_MIXED)
        return fail(nullptr, OMOK_ERR_INVALID, "bad net_mode");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, OMOK_ERR_HIP, "no HIP device available: this library has no CPU path");
    if (cfg->device < 0 || cfg->device >= ndev) return fail(nullptr, OMOK_ERR_INVALID, "device %d out of range (%d devices)", cfg->device, ndev);
    { // the re-rooting kernel keeps 3 B per node + 2 B per table in LDS: refuse arenas it could not launch with
        hipDeviceProp_t prop;
        const size_t lds = advance_lds_bytes(cfg->max_nodes, cfg->max_tables);
        if (hipGetDeviceProperties(&prop, cfg->device) != hipSuccess) return fail(nullptr, OMOK_ERR_HIP, "hipGetDeviceProperties failed");
        const size_t limit = prop.sharedMemPerBlockOptin > prop.sharedMemPerBlock ? prop.sharedMemPerBlockOptin : prop.sharedMemPerBlock;
        if (lds > limit)
            return fail(nullptr, OMOK_ERR_INVALID, "max_nodes=%d / max_tables=%d need %zu bytes of LDS in the re-rooting kernel, the device allows %zu",
                        cfg->max_nodes, cfg->max_tables, lds, limit);
    }
    omok_engine* e = new omok_engine();
    e->cfg = *cfg;
    e->n = cfg->board_size;
    e->hw = e->n * e->n;
    e->rowp = (e->hw + 63) / 64 * 64;
    e->nw = (e->hw + 63) / 64;
    e->T = 2 * cfg->games;
    hipError_t r = hipSetDevice(cfg->device);
    if (r == hipSuccess) r = hipStreamCreateWithFlags(&e->st, hipStreamNonBlocking);
    if (r != hipSuccess) {
        g_create_error = std::string("HIP init failed: ") + hipGetErrorString(r);
        delete e;
        return OMOK_ERR_HIP;
    }
    Store& S = e->S;
    S.cap_nodes = cfg->max_nodes;
    S.cap_tables = cfg->max_tables;
    S.stride_nodes = cfg->max_nodes | 1;   // odd strides: consecutive trees on different HBM channels (common.h, Store)
    S.stride_tables = cfg->max_tables | 1;
    S.games = cfg->games;
    const size_t T = (size_t)e->T, cn = (size_t)S.stride_nodes, ct = (size_t)S.stride_tables, rp = (size_t)e->rowp;
    const size_t G = (size_t)cfg->games, HW = (size_t)e->hw;
    if (cfg->max_tree_waves < 0 || cfg->max_tree_waves > MAX_TREE_WAVES)
        { omok_destroy(e); return fail(nullptr, OMOK_ERR_INVALID, "max_tree_waves must be in [0, %d]", MAX_TREE_WAVES); }
    const size_t max_b = std::max(G, (size_t)cfg->max_tree_waves) * (size_t)cfg->max_batch_k;
    int rc = 0;
    rc |= dalloc(e, &S.hdr, T * cn);
    rc |= dalloc(e, &S.board, T * cn * 2 * e->nw);
    rc |= dalloc(e, &S.policy, T * cn * rp);
    rc |= dalloc(e, &S.tcn, T * ct * rp);
    rc |= dalloc(e, &S.tcw, T * ct * rp);
    rc |= dalloc(e, &S.tcidx, T * ct * rp);
    rc |= dalloc(e, &S.tcorder, T * ct * rp);
    rc |= dalloc(e, &S.towner, T * ct);
    rc |= dalloc(e, &S.ts, T);
    rc |= dalloc(e, &S.req_node, T * KMAX);
    rc |= dalloc(e, &S.gs, G);
    rc |= dalloc(e, &S.req_ref, max_b);
    rc |= dalloc(e, &S.req_aux, max_b);
    rc |= dalloc(e, &S.d_count, 4);
    rc |= dalloc(e, &S.rp_board, G * HW * 2 * e->nw);
    rc |= dalloc(e, &S.rp_turn, G * HW);
    rc |= dalloc(e, &S.rp_pi, G * HW * rp);
    rc |= dalloc(e, &S.rp_z, G * HW);
    rc |= dalloc(e, &S.d_bytes, 2);
    rc |= dalloc(e, &e->d_root_policy, rp);
    rc |= dalloc(e, &e->d_actions, G);
    rc |= dalloc(e, &e->d_error, 4);
    rc |= dalloc(e, &e->d_evals, 2);
    rc |= dalloc(e, &e->d_sh_req, (size_t)MAX_TREE_WAVES * KMAX);
    rc |= dalloc(e, &e->d_sh_cnt, 2 * (size_t)KMAX);
    rc |= dalloc(e, &e->d_flags, 4);
    rc |= dalloc(e, &e->d_pi, G * HW);
    rc |= dalloc(e, &e->d_has, G);
    rc |= dalloc(e, &e->d_aug_offsets, G + 1);
    if (rc) {
        g_create_error = e->err;
        omok_destroy(e);
        return OMOK_ERR_HIP;
    }
    hipMemsetAsync(S.d_count, 0, 16, e->st);
    hipMemsetAsync(S.d_bytes, 0, 16, e->st);
    hipMemsetAsync(e->d_error, 0, 16, e->st);
    hipMemsetAsync(e->d_evals, 0, 16, e->st);
    hipMemsetAsync(S.gs, 0, sizeof(GameState) * G, e->st);
    hipMemsetAsync(S.ts, 0, sizeof(TreeState) * T, e->st);
    if (set_advance_lds_attribute(e->n, advance_lds_bytes(cfg->max_nodes, cfg->max_tables)) != 0) {
        g_create_error = "hipFuncSetAttribute(k_advance, max dynamic LDS) failed";
        omok_destroy(e);
        return OMOK_ERR_HIP;
    }
    e->net.device = cfg->device;
    e->net.n = e->n;
    e->net.hw = e->hw;
    e->net.rowp = e->rowp;
    e->net.mode = cfg->net_mode == OMOK_NET_F32 ? OMOK_NET_F32 : OMOK_NET_F16X3;
    e->net.siblings = cfg->net_mode != OMOK_NET_F16X3_ROWS;
    e->net.fc0_policy = cfg->net_mode == OMOK_NET_F16X3_FP6 ? FC0_FP6 : cfg->net_mode == OMOK_NET_F16X3_F16 ? FC0_F16
                      : cfg->net_mode == OMOK_NET_F16X3_MIXED ? FC0_MIXED : FC0_AUTO;
    e->net.max_b = (int)max_b;
    e->net.games = cfg->games;
    for (int i = 0; i < NET_TENSORS; ++i) e->net.wsize[i] = net_tensor_size(e->n, i);
    if (net_alloc(e->net) == 0) {
        g_create_error = "net buffer allocation failed (hipMalloc)";
        omok_destroy(e);
        return OMOK_ERR_HIP;
    }
    if (hipStreamSynchronize(e->st) != hipSuccess) {
        g_create_error = "stream sync failed after init";
        omok_destroy(e);
        return OMOK_ERR_HIP;
    }
    *out = e;
    return OMOK_OK;
}

// ---- net ---------------------------------------------------------------------------------------
extern "C" int omok_net_num_tensors(void) { return NET_TENSORS; }
extern "C" int64_t omok_net_tensor_size(const omok_engine* e, int index) { return e ? net_tensor_size(e->n, index) : -1; }

extern "C" int omok_net_load(omok_engine* e, int index, const float* data, int64_t count) {
    if (!e || !data) return OMOK_ERR_INVALID;
    if (index < 0 || index >= NET_TENSORS) return fail(e, OMOK_ERR_INVALID, "tensor index %d out of range", index);
    if (count != e->net.wsize[index]) return fail(e, OMOK_ERR_INVALID, "tensor %d: expected %lld values, got %lld", index, (long long)e->net.wsize[index], (long long)count);
    HIPCHK(e, hipSetDevice(e->cfg.device));
    HIPCHK(e, hipMemcpyAsync(e->net.w[index], data, sizeof(float) * (size_t)count, hipMemcpyHostToDevice, e->st));
    HIPCHK(e, hipStreamSynchronize(e->st));
    e->net.loaded[index] = true;
    e->net.committed = false;
    return OMOK_OK;
}

extern "C" int omok_net_commit(omok_engine* e) {
    if (!e) return OMOK_ERR_INVALID;
    for (int i = 0; i < NET_TENSORS; ++i)
        if (!e->net.loaded[i]) return fail(e, OMOK_ERR_STATE, "tensor %d was never loaded", i);
    HIPCHK(e, hipSetDevice(e->cfg.device));
    if (e->round_reqs >= 0 || e->mirror_reqs >= 0) return fail(e, OMOK_ERR_STATE, "omok_net_commit while a round / mirror batch is pending");
    if (net_commit(e->net, e->S, e->st) != 0) return fail(e, OMOK_ERR_HIP, "weight packing / format probe failed");
    net_invalidate_sibling_cache(e->net); // (new weights: cached base evaluations are void)
    HIPCHK(e, hipStreamSynchronize(e->st));
    e->net.committed = true;
    return OMOK_OK;
}

// ---- weights file: ModelIO::save / ModelIO::load (alpha-zero/src/model_io.rs:20-24,59-120) ---------------------
// bincode 1.3.3 default options: little-endian, fixed-width integers; Vec<T> = u64 length + elements; String = u64
// byte length + UTF-8 bytes.  SavedData = { variable_names: Vec<String>, parameters: Vec<Vec<f32>> }.
static const char* net_tensor_name(int i, char* buf, size_t cap) {
    static const char* head[2] = {"conv_w", "conv_b"};
    static const char* blk[7] = {"conv0_w", "conv0_b", "conv1_w(depthwise)", "conv1_w(pointwise)", "conv1_b", "conv2_w", "conv2_b"};
    static const char* tail[8] = {"fc0_w", "fc0_b", "fc1_w", "fc1_b", "v_fc0_w", "v_fc0_b", "p_fc0_w", "p_fc0_b"};
    if (i < 2) snprintf(buf, cap, "%s", head[i]);
    else if (i < 23) snprintf(buf, cap, "residual_%d_%s", (i - 2) / 7, blk[(i - 2) % 7]);
    else snprintf(buf, cap, "%s", tail[i - 23]);
    return buf;
}

static bool rd_u64(FILE* f, uint64_t* v) {
    unsigned char b[8];
    if (fread(b, 1, 8, f) != 8) return false;
    uint64_t x = 0;
    for (int i = 7; i >= 0; --i) x = (x << 8) | b[i];
    *v = x;
    return true;
}
static bool wr_u64(FILE* f, uint64_t v) {
    unsigned char b[8];
    for (int i = 0; i < 8; ++i) b[i] = (unsigned char)(v >> (8 * i));
    return fwrite(b, 1, 8, f) == 8;
}

extern "C" int omok_net_load_file(omok_engine* e, const char* path) {
    if (!e || !path) return OMOK_ERR_INVALID;
    FILE* f = fopen(path, "rb");
    if (!f) return fail(e, OMOK_ERR_INVALID, "cannot open weights file %s", path);
    fseek(f, 0, SEEK_END);
    const long long fsize = ftell(f);
    fseek(f, 0, SEEK_SET);
    int rc = OMOK_OK;
    uint64_t n_names = 0, n_params = 0;
    std::vector<std::vector<float>> params(NET_TENSORS); // validated completely before the engine is touched
    if (!rd_u64(f, &n_names) || (long long)n_names > fsize / 8) rc = fail(e, OMOK_ERR_INVALID, "weights file %s: bad name count", path);
    for (uint64_t i = 0; rc == OMOK_OK && i < n_names; ++i) { // names are skipped: the load is positional (model_io.rs:98)
        uint64_t len = 0;
        if (!rd_u64(f, &len) || (long long)len > fsize || fseek(f, (long)len, SEEK_CUR) != 0) rc = fail(e, OMOK_ERR_INVALID, "weights file %s: truncated in variable_names", path);
    }
    if (rc == OMOK_OK && (!rd_u64(f, &n_params) || (long long)n_params > fsize / 8)) rc = fail(e, OMOK_ERR_INVALID, "weights file %s: bad parameter count", path);
    if (rc == OMOK_OK && n_params < (uint64_t)NET_TENSORS)
        rc = fail(e, OMOK_ERR_INVALID, "weights file %s holds %llu parameter vectors, the net has %d variables", path, (unsigned long long)n_params, NET_TENSORS);
    for (int i = 0; rc == OMOK_OK && i < NET_TENSORS; ++i) {
        uint64_t len = 0;
        if (!rd_u64(f, &len)) { rc = fail(e, OMOK_ERR_INVALID, "weights file %s: truncated at parameter %d", path, i); break; }
        if ((int64_t)len != e->net.wsize[i]) { // Tensor::copy_from_slice panics on a length mismatch (model_io.rs:106)
            rc = fail(e, OMOK_ERR_INVALID, "weights file %s: parameter %d has %llu values, variable has %lld", path, i, (unsigned long long)len, (long long)e->net.wsize[i]);
            break;
        }
        params[i].resize((size_t)len); // (host is little-endian like the file)
        if (fread(params[i].data(), sizeof(float), (size_t)len, f) != (size_t)len) { rc = fail(e, OMOK_ERR_INVALID, "weights file %s: truncated inside parameter %d", path, i); break; }
    }
    fclose(f);
    for (int i = 0; rc == OMOK_OK && i < NET_TENSORS; ++i) rc = omok_net_load(e, i, params[i].data(), (int64_t)params[i].size());
    if (rc != OMOK_OK) return rc;
    return omok_net_commit(e);
}

extern "C" int omok_net_save_file(omok_engine* e, const char* path) {
    if (!e || !path) return OMOK_ERR_INVALID;
    for (int i = 0; i < NET_TENSORS; ++i)
        if (!e->net.loaded[i]) return fail(e, OMOK_ERR_STATE, "tensor %d was never loaded", i);
    HIPCHK(e, hipSetDevice(e->cfg.device));
    FILE* f = fopen(path, "wb");
    if (!f) return fail(e, OMOK_ERR_INVALID, "cannot create weights file %s", path);
    bool ok = wr_u64(f, NET_TENSORS);
    char name[64];
    for (int i = 0; ok && i < NET_TENSORS; ++i) {
        net_tensor_name(i, name, sizeof(name));
        const size_t len = strlen(name);
        ok = wr_u64(f, len) && fwrite(name, 1, len, f) == len;
    }
    ok = ok && wr_u64(f, NET_TENSORS);
    std::vector<float> buf;
    for (int i = 0; ok && i < NET_TENSORS; ++i) {
        const size_t len = (size_t)e->net.wsize[i];
        buf.resize(len);
        if (hipMemcpy(buf.data(), e->net.w[i], sizeof(float) * len, hipMemcpyDeviceToHost) != hipSuccess) { fclose(f); return fail(e, OMOK_ERR_HIP, "weight read-back failed"); }
        ok = wr_u64(f, len) && fwrite(buf.data(), sizeof(float), len, f) == len;
    }
    ok = (fclose(f) == 0) && ok;
    return ok ? OMOK_OK : fail(e, OMOK_ERR_INVALID, "short write to %s", path);
}

static int need_net(omok_engine* e) {
    if (!e->net.committed) return fail(e, OMOK_ERR_STATE, "net not loaded/committed (omok_net_load x31 + omok_net_commit)");
    return 0;
}

static int check_async(omok_engine* e, const char* what) {
    hipError_t r = hipGetLastError();
    if (r != hipSuccess) return fail(e, OMOK_ERR_HIP, "%s: launch failed: %s", what, hipGetErrorString(r));
    return 0;
}

static int sync_and_check(omok_engine* e, const char* what) {
    hipError_t r = hipStreamSynchronize(e->st);
    if (r != hipSuccess) return fail(e, OMOK_ERR_HIP, "%s: %s", what, hipGetErrorString(r));
    return check_async(e, what);
}

extern "C" int omok_evaluate_pv(omok_engine* e, const float* in, int32_t batch, float* p, float* v) {
    if (!e || !in || !p || batch < 0) return OMOK_ERR_INVALID;
    if (need_net(e)) return OMOK_ERR_STATE;
    if (e->round_reqs >= 0 || e->mirror_reqs >= 0) // the forward reuses the request counter and the output rows of the pending batch
        return fail(e, OMOK_ERR_STATE, "omok_evaluate_pv between omok_round_generate / omok_mirror_generate and their scatter / apply");
    HIPCHK(e, hipSetDevice(e->cfg.device));
    const size_t in_row = 3 * (size_t)e->hw;
    float* d_pack = e->net.in_f32; // reused as the packed output staging after the forward
    for (int32_t done = 0; done < batch; done += e->net.max_b) {
        const int b = std::min<int32_t>(e->net.max_b, batch - done);
        HIPCHK(e, hipMemcpyAsync(e->net.in_f32, in + (size_t)done * in_row, sizeof(float) * in_row * b, hipMemcpyHostToDevice, e->st));
        HIPCHK(e, hipMemcpyAsync(e->S.d_count, &b, sizeof(int32_t), hipMemcpyHostToDevice, e->st));
        net_forward_inputs(e->net, e->S, b, e->st, &e->prof);
        const size_t tot = (size_t)b * e->hw;
        k_pack_rows<<<(unsigned)((tot + 255) / 256), 256, 0, e->st>>>(e->net.p, d_pack, e->hw, e->rowp, b);
        HIPCHK(e, hipMemcpyAsync(p + (size_t)done * e->hw, d_pack, sizeof(float) * tot, hipMemcpyDeviceToHost, e->st));
        if (v) HIPCHK(e, hipMemcpyAsync(v + done, e->net.v, sizeof(float) * b, hipMemcpyDeviceToHost, e->st));
        if (sync_and_check(e, "evaluate_pv")) return OMOK_ERR_HIP;
        e->evals += b;
    }
    return OMOK_OK;
}

// Debug / evidence entry point: the same forward, but the PRE-softmax policy logits and the PRE-tanh value (network.rs:188-247
// before the Tanh / Softmax ops), so that precision can be stated on logits as well as on the outputs the reference API returns.
extern "C" int omok_evaluate_logits(omok_engine* e, const float* in, int32_t batch, float* logits, float* vpre) {
    if (!e || !in || !logits || batch < 0) return OMOK_ERR_INVALID;
    if (need_net(e)) return OMOK_ERR_STATE;
    if (e->round_reqs >= 0 || e->mirror_reqs >= 0) return fail(e, OMOK_ERR_STATE, "omok_evaluate_logits while a round / mirror batch is pending");
    HIPCHK(e, hipSetDevice(e->cfg.device));
    const size_t in_row = 3 * (size_t)e->hw;
    const int piece = e->net.mode == OMOK_NET_F32 ? std::min(e->net.chunk, e->net.max_b) : e->net.max_b;
    float* d_pack = e->net.in_f32;
    for (int32_t done = 0; done < batch; done += piece) {
        const int b = std::min<int32_t>(piece, batch - done);
        HIPCHK(e, hipMemcpyAsync(e->net.in_f32, in + (size_t)done * in_row, sizeof(float) * in_row * b, hipMemcpyHostToDevice, e->st));
        HIPCHK(e, hipMemcpyAsync(e->S.d_count, &b, sizeof(int32_t), hipMemcpyHostToDevice, e->st));
        net_forward_inputs(e->net, e->S, b, e->st, &e->prof);
        int stride = 0;
        const float* lg = net_logits(e->net, &stride);
        const size_t tot = (size_t)b * e->hw;
        k_pack_rows<<<(unsigned)((tot + 255) / 256), 256, 0, e->st>>>(lg, d_pack, e->hw, stride, b);
        HIPCHK(e, hipMemcpyAsync(logits + (size_t)done * e->hw, d_pack, sizeof(float) * tot, hipMemcpyDeviceToHost, e->st));
        if (vpre) HIPCHK(e, hipMemcpyAsync(vpre + done, e->net.vpre, sizeof(float) * b, hipMemcpyDeviceToHost, e->st));
        if (sync_and_check(e, "evaluate_logits")) return OMOK_ERR_HIP;
        e->evals += b;
    }
    return OMOK_OK;
}

// ---- environment -------------------------------------------------------------------------------
extern "C" int omok_env_play(omok_engine* e, const int32_t* moves, int32_t batch, int32_t len, int32_t* status_out,
                             uint8_t* boards_out, uint8_t* turns_out, uint16_t* legal_out) {
    if (!e || batch < 1 || len < 0 || (len > 0 && !moves)) return OMOK_ERR_INVALID;
    HIPCHK(e, hipSetDevice(e->cfg.device));
    int32_t *d_moves = nullptr, *d_status = nullptr;
    uint8_t *d_boards = nullptr, *d_turns = nullptr;
    uint16_t* d_legal = nullptr;
    const size_t ml = (size_t)batch * (size_t)(len > 0 ? len : 1);
    HIPCHK(e, hipMalloc((void**)&d_moves, ml * 4));
    HIPCHK(e, hipMalloc((void**)&d_status, ml * 4));
    HIPCHK(e, hipMalloc((void**)&d_boards, (size_t)batch * e->hw));
    HIPCHK(e, hipMalloc((void**)&d_turns, (size_t)batch));
    HIPCHK(e, hipMalloc((void**)&d_legal, (size_t)batch * 2));
    if (len > 0) hipMemcpyAsync(d_moves, moves, ml * 4, hipMemcpyHostToDevice, e->st);
    launch_env_play(e->n, d_moves, batch, len, d_status, d_boards, d_turns, d_legal, e->st);
    if (status_out && len > 0) hipMemcpyAsync(status_out, d_status, ml * 4, hipMemcpyDeviceToHost, e->st);
    if (boards_out) hipMemcpyAsync(boards_out, d_boards, (size_t)batch * e->hw, hipMemcpyDeviceToHost, e->st);
    if (turns_out) hipMemcpyAsync(turns_out, d_turns, (size_t)batch, hipMemcpyDeviceToHost, e->st);
    if (legal_out) hipMemcpyAsync(legal_out, d_legal, (size_t)batch * 2, hipMemcpyDeviceToHost, e->st);
    const int rc = sync_and_check(e, "env_play");
    hipFree(d_moves); hipFree(d_status); hipFree(d_boards); hipFree(d_turns); hipFree(d_legal);
    return rc ? OMOK_ERR_HIP : OMOK_OK;
}

extern "C" int omok_encode_nn_input(omok_engine* e, const uint8_t* boards, const uint8_t* turns, int32_t batch,
                                    int32_t mode, float* out) {
    if (!e || !boards || !turns || !out || batch < 1 || (mode != 0 && mode != 1)) return OMOK_ERR_INVALID;
    HIPCHK(e, hipSetDevice(e->cfg.device));
    uint8_t *d_boards = nullptr, *d_turns = nullptr;
    float* d_out = nullptr;
    HIPCHK(e, hipMalloc((void**)&d_boards, (size_t)batch * e->hw));
    HIPCHK(e, hipMalloc((void**)&d_turns, (size_t)batch));
    HIPCHK(e, hipMalloc((void**)&d_out, (size_t)batch * 3 * e->hw * 4));
    hipMemcpyAsync(d_boards, boards, (size_t)batch * e->hw, hipMemcpyHostToDevice, e->st);
    hipMemcpyAsync(d_turns, turns, (size_t)batch, hipMemcpyHostToDevice, e->st);
    launch_encode_boards(e->n, d_boards, d_turns, batch, mode, d_out, e->st);
    hipMemcpyAsync(out, d_out, (size_t)batch * 3 * e->hw * 4, hipMemcpyDeviceToHost, e->st);
    const int rc = sync_and_check(e, "encode_nn_input");
    hipFree(d_boards); hipFree(d_turns); hipFree(d_out);
    return rc ? OMOK_ERR_HIP : OMOK_OK;
}

// ---- self-play ---------------------------------------------------------------------------------
static int read_status(omok_engine* e, uint32_t* err_bits, uint32_t* alive) {
    k_count_alive<<<1, 1024, 0, e->st>>>(e->S, e->d_error, e->d_evals);
    uint32_t h[4] = {0, 0, 0, 0};
    HIPCHK(e, hipMemcpyAsync(h, e->d_error, 16, hipMemcpyDeviceToHost, e->st));
    if (sync_and_check(e, "status readback")) return OMOK_ERR_HIP;
    if (err_bits) *err_bits = h[0];
    if (alive) *alive = h[1];
    if (h[2] > e->peak_nodes) e->peak_nodes = h[2];
    if (h[3] > e->peak_tables) e->peak_tables = h[3];
    return 0;
}

static int tree_error(omok_engine* e, uint32_t bits) {
    if (bits & 1u) return fail(e, OMOK_ERR_OVERFLOW, "a tree arena overflowed (max_nodes=%d, max_tables=%d; peak use %u nodes, %u tables): raise them", e->cfg.max_nodes, e->cfg.max_tables, e->peak_nodes, e->peak_tables);
    if (bits & 4u) return fail(e, OMOK_ERR_ILLEGAL, "sample_action on a tree with no visited children (run execute first)");
    if (bits) return fail(e, OMOK_ERR_ILLEGAL, "illegal tree operation (error bits 0x%x)", bits);
    return 0;
}

extern "C" int omok_selfplay_reset(omok_engine* e) {
    if (!e) return OMOK_ERR_INVALID;
    if (need_net(e)) return OMOK_ERR_STATE;
    HIPCHK(e, hipSetDevice(e->cfg.device));
    // Agent::new: evaluate_p on the empty board in Player mode (agent.rs:19-20); the result is the
    // same for every tree of a fixed net, so it is computed once.
    std::vector<float> in(3 * (size_t)e->hw, 0.0f);
    for (int i = 2 * e->hw; i < 3 * e->hw; ++i) in[i] = 1.0f; // Black to move (encoder.rs:34-37)
    const int one = 1;
    HIPCHK(e, hipMemcpyAsync(e->net.in_f32, in.data(), sizeof(float) * in.size(), hipMemcpyHostToDevice, e->st));
    HIPCHK(e, hipMemcpyAsync(e->S.d_count, &one, sizeof(int32_t), hipMemcpyHostToDevice, e->st));
    net_forward_inputs(e->net, e->S, 1, e->st, &e->prof);
    k_copy_root_policy<<<1, 256, 0, e->st>>>(e->net.p, e->d_root_policy, e->hw, e->rowp);
    launch_reset(e->n, e->S, e->d_root_policy, e->st);
    net_invalidate_sibling_cache(e->net);
    if (sync_and_check(e, "selfplay_reset")) return OMOK_ERR_HIP;
    e->evals += 1;
    e->key = e->cfg.seed + e->episode * 0x9E3779B97F4A7C15ULL; // a fresh RNG stream per episode (the reference draws thread_rng anew, trainer.rs:71-93)
    e->episode += 1;
    e->ply = 0;
    e->reset_done = true;
    e->sampled = false;
    e->round_reqs = e->mirror_reqs = -1;
    return OMOK_OK;
}

static int need_reset(omok_engine* e) {
    if (!e->reset_done) return fail(e, OMOK_ERR_STATE, "omok_selfplay_reset has not been called");
    return 0;
}

// defer_backups (run loops): this round's backups are not launched here; pending_backups: the previous round's run at the head of this round's
// kernel.  The caller launches the last round's backups itself (launch_backups).
static void enqueue_round(omok_engine* e, int round, int K, float eps, float alpha, bool eval_and_scatter, int alive, bool defer_backups = false,
                          bool pending_backups = false) {
    const int side = e->ply & 1;
    RoundArgs a{side, round, K, e->ply, eps, alpha, e->key, e->cfg.game_offset, pending_backups ? e->net.v : nullptr};
    e->prof.round_begin();
    e->prof.begin(PC_ROUND, e->st);
    launch_round(e->n, e->S, a, e->st);
    e->prof.end(e->st);
    e->prof.begin(PC_TREE_OTHER, e->st);
    // run-loop rounds whose forward groups the requests by parent: k_scan zeroes the grouping counters, k_group writes the dense request list
    // (two launches less per round); the step-wise API keeps the separate kernels (its callers read the request list before the forward)
    int max_req = alive * K;
    if (max_req > e->net.max_b) max_req = e->net.max_b;
    const bool sib_round = eval_and_scatter && net_round_takes_sibling_path(e->net, max_req);
    launch_scan(e->n, e->S, side, K, e->st, e->d_evals, sib_round ? e->net.d_gcnt : nullptr, NET_GCNT_INTS, !sib_round);
    e->net.gcnt_zeroed = e->net.fill_in_group = sib_round;
    e->net.fill_side = side;
    e->net.fill_k = K;
    e->prof.end(e->st);
    if (eval_and_scatter) {
        // the split-precision net hands over its logits: softmax / tanh run inside the policy scatter (no [requests][ROWP] round trip of p)
        const bool fused = net_logits_cover_batch(e->net, alive * K);
        net_forward_requests(e->net, e->S, alive * K, e->st, &e->prof, side, fused);
        e->prof.begin(PC_TREE_OTHER, e->st);
        if (fused) {
            int lrow = 0;
            const float* lg = net_logits(e->net, &lrow);
            launch_softmax_scatter(e->n, e->S, side, lg, lrow, e->net.v, e->net.vpre, alive * K, e->st, !defer_backups);
        } else launch_scatter(e->n, e->S, side, e->net.p, e->net.v, alive * K, e->st, !defer_backups);
        e->prof.end(e->st);
    }
    e->prof.round_end();
}

static int enqueue_execute(omok_engine* e, int count, int K, float eps, float alpha, int alive) {
    int processed = 0, round = 0;
    while (processed < count) { // pme.rs:39-42,207
        enqueue_round(e, round, K, eps, alpha, true, alive, true, round > 0);
        processed += K;
        round += 1;
    }
    if (round > 0) { // the last round's backups
        e->prof.begin(PC_TREE_OTHER, e->st);
        launch_backups(e->n, e->S, e->ply & 1, e->net.v, e->st);
        e->prof.end(e->st);
    }
    return round;
}

static int check_exec_args(omok_engine* e, int count, int K, float eps, float alpha) {
    if (count < 1) return fail(e, OMOK_ERR_INVALID, "count must be >= 1");
    if (K < 1 || K > e->cfg.max_batch_k) return fail(e, OMOK_ERR_INVALID, "batch_size %d outside [1, max_batch_k=%d]", K, e->cfg.max_batch_k);
    if (!(alpha > 0.0f)) return fail(e, OMOK_ERR_INVALID, "alpha must be > 0");
    if (!(eps >= 0.0f && eps <= 1.0f)) return fail(e, OMOK_ERR_INVALID, "epsilon must be in [0,1]");
    return 0;
}

extern "C" int omok_execute(omok_engine* e, int32_t count, int32_t batch_size, float epsilon, float alpha) {
    if (!e) return OMOK_ERR_INVALID;
    if (need_net(e) || need_reset(e)) return OMOK_ERR_STATE;
    if (check_exec_args(e, count, batch_size, epsilon, alpha)) return OMOK_ERR_INVALID;
    HIPCHK(e, hipSetDevice(e->cfg.device));
    uint32_t bits = 0, alive = 0;
    if (read_status(e, &bits, &alive)) return OMOK_ERR_HIP;
    const int rounds = enqueue_execute(e, count, batch_size, epsilon, alpha, (int)alive);
    e->sims += (double)rounds * batch_size * alive;
    if (read_status(e, &bits, &alive)) return OMOK_ERR_HIP;
    return tree_error(e, bits);
}

// MCTSExecutor::run (alpha-zero/src/mcts_executor.rs:29-255): ONE tree, rounds as concurrent tasks.  `waves` rounds run at a time
// on the shared tree of game 0 (one workgroup of `waves` wavefronts); their requests form one net batch.
extern "C" int omok_execute_shared(omok_engine* e, int32_t count, int32_t batch_size, float epsilon, float alpha, int32_t waves) {
    if (!e) return OMOK_ERR_INVALID;
    if (need_net(e) || need_reset(e)) return OMOK_ERR_STATE;
    if (check_exec_args(e, count, batch_size, epsilon, alpha)) return OMOK_ERR_INVALID;
    if (e->cfg.games != 1) return fail(e, OMOK_ERR_INVALID, "omok_execute_shared searches ONE tree: create the engine with games = 1 (got %d)", e->cfg.games);
    if (waves < 1 || waves > MAX_TREE_WAVES || waves * batch_size > e->net.max_b) // (cfg.max_tree_waves sizes the net batch: it binds through max_b)
        return fail(e, OMOK_ERR_INVALID, "waves must be in [1, %d] and waves * batch_size <= %d (the net batch, sized by max(games, max_tree_waves = %d) * max_batch_k)",
                    MAX_TREE_WAVES, e->net.max_b, e->cfg.max_tree_waves);
    HIPCHK(e, hipSetDevice(e->cfg.device));
    uint32_t bits = 0, alive = 0;
    if (read_status(e, &bits, &alive)) return OMOK_ERR_HIP;
    const int side = e->ply & 1;
    RoundArgs a{side, 0, 0, e->ply, epsilon, alpha, e->key, e->cfg.game_offset};
    e->prof.begin(PC_ROUND, e->st);
    launch_round(e->n, e->S, a, e->st); // K = 0, round 0: the root's Dirichlet noise only (mcts_executor.rs:38-68)
    e->prof.end(e->st);
    int exec_count = count / batch_size; // :70-74
    if (exec_count * batch_size != count) exec_count += 1;
    a.K = batch_size;
    for (int group = 0; group * waves < exec_count; ++group) {
        e->prof.begin(PC_ROUND, e->st);
        launch_round_shared(e->n, e->S, a, exec_count, group, waves, e->d_sh_req, e->d_sh_cnt, e->st);
        k_add_evals<<<1, 64, 0, e->st>>>(e->S.d_count, e->d_evals);
        e->prof.end(e->st);
        net_forward_requests(e->net, e->S, waves * batch_size, e->st, &e->prof);
        e->prof.begin(PC_TREE_OTHER, e->st);
        launch_scatter_shared(e->n, e->S, side, e->net.p, e->net.v, waves * batch_size, waves, e->d_sh_req, e->d_sh_cnt, e->st);
        e->prof.end(e->st);
    }
    e->sims += (double)exec_count * batch_size * alive;
    if (read_status(e, &bits, &alive)) return OMOK_ERR_HIP;
    return tree_error(e, bits);
}

// The same search with a RECORDED interleaving (tests): every simulation and every backup of the scatter phase runs under the tree lock and
// the lock order is written down, group by group, together with the net outputs of the group's requests -- everything a CPU restatement of
// MCTSExecutor::run needs to replay the run exactly (tests/test_gpu_shared_tree.py).
extern "C" int omok_execute_shared_recorded(omok_engine* e, int32_t count, int32_t batch_size, float epsilon, float alpha, int32_t waves,
                                            uint8_t* sim_order, uint8_t* backup_order, int32_t* group_counts, float* p, float* v,
                                            int32_t cap_requests, int32_t* n_groups, int32_t* n_requests) {
    if (!e || !sim_order || !backup_order || !group_counts || !p || !v || !n_groups || !n_requests) return OMOK_ERR_INVALID;
    if (need_net(e) || need_reset(e)) return OMOK_ERR_STATE;
    if (check_exec_args(e, count, batch_size, epsilon, alpha)) return OMOK_ERR_INVALID;
    if (e->cfg.games != 1) return fail(e, OMOK_ERR_INVALID, "omok_execute_shared_recorded searches ONE tree: create the engine with games = 1 (got %d)", e->cfg.games);
    if (waves < 1 || waves > MAX_TREE_WAVES || waves * batch_size > e->net.max_b) // (cfg.max_tree_waves sizes the net batch: it binds through max_b)
        return fail(e, OMOK_ERR_INVALID, "waves must be in [1, %d] and waves * batch_size <= %d (the net batch, sized by max(games, max_tree_waves = %d) * max_batch_k)",
                    MAX_TREE_WAVES, e->net.max_b, e->cfg.max_tree_waves);
    ENTER(e);
    const size_t per = (size_t)waves * batch_size;
    uint8_t* d_rec = nullptr;   // [2][per] wave ids: simulations, backups
    uint32_t* d_pos = nullptr;  // [2]
    if (hipMalloc((void**)&d_rec, 2 * per) != hipSuccess || hipMalloc((void**)&d_pos, 8) != hipSuccess) {
        if (d_rec) hipFree(d_rec);
        return fail(e, OMOK_ERR_HIP, "recorded mode: device allocation failed");
    }
    auto done = [&](int rc) { hipFree(d_rec); hipFree(d_pos); return rc; };
    uint32_t bits = 0, alive = 0;
    if (read_status(e, &bits, &alive)) return done(OMOK_ERR_HIP);
    const int side = e->ply & 1;
    RoundArgs a{side, 0, 0, e->ply, epsilon, alpha, e->key, e->cfg.game_offset};
    launch_round(e->n, e->S, a, e->st); // K = 0, round 0: the root's Dirichlet noise only (mcts_executor.rs:38-68)
    int exec_count = count / batch_size; // :70-74
    if (exec_count * batch_size != count) exec_count += 1;
    a.K = batch_size;
    int groups = 0, total_req = 0;
    std::vector<float> prow((size_t)per * e->rowp);
    for (int group = 0; group * waves < exec_count; ++group, ++groups) {
        hipMemsetAsync(d_rec, 0xFF, 2 * per, e->st);
        hipMemsetAsync(d_pos, 0, 8, e->st);
        launch_round_shared(e->n, e->S, a, exec_count, group, waves, e->d_sh_req, e->d_sh_cnt, e->st, d_rec, d_pos);
        k_add_evals<<<1, 64, 0, e->st>>>(e->S.d_count, e->d_evals);
        int32_t cnt = 0;
        if (hipMemcpyAsync(&cnt, e->S.d_count, 4, hipMemcpyDeviceToHost, e->st) != hipSuccess || sync_and_check(e, "execute_shared_recorded")) return done(OMOK_ERR_HIP);
        if (total_req + cnt > cap_requests) return done(fail(e, OMOK_ERR_INVALID, "recorded mode: more than cap_requests = %d requests", cap_requests));
        if (cnt > 0) {
            net_forward_requests(e->net, e->S, waves * batch_size, e->st, &e->prof);
            hipMemcpyAsync(prow.data(), e->net.p, sizeof(float) * (size_t)cnt * e->rowp, hipMemcpyDeviceToHost, e->st);
            hipMemcpyAsync(v + total_req, e->net.v, sizeof(float) * cnt, hipMemcpyDeviceToHost, e->st);
        }
        launch_scatter_shared(e->n, e->S, side, e->net.p, e->net.v, waves * batch_size, waves, e->d_sh_req, e->d_sh_cnt, e->st, d_rec + per, d_pos + 1);
        uint32_t pos[2] = {0, 0};
        hipMemcpyAsync(sim_order + (size_t)group * per, d_rec, per, hipMemcpyDeviceToHost, e->st);
        hipMemcpyAsync(backup_order + (size_t)group * per, d_rec + per, per, hipMemcpyDeviceToHost, e->st);
        hipMemcpyAsync(pos, d_pos, 8, hipMemcpyDeviceToHost, e->st);
        if (sync_and_check(e, "execute_shared_recorded")) return done(OMOK_ERR_HIP);
        for (int r = 0; r < cnt; ++r) memcpy(p + (size_t)(total_req + r) * e->hw, prow.data() + (size_t)r * e->rowp, sizeof(float) * e->hw);
        group_counts[3 * group] = (int32_t)pos[0];
        group_counts[3 * group + 1] = (int32_t)pos[1];
        group_counts[3 * group + 2] = cnt;
        total_req += cnt;
    }
    *n_groups = groups;
    *n_requests = total_req;
    e->sims += (double)exec_count * batch_size * alive;
    if (read_status(e, &bits, &alive)) return done(OMOK_ERR_HIP);
    return done(tree_error(e, bits));
}

static void enqueue_sample(omok_engine* e, float temperature, int threshold) {
    e->prof.begin(PC_PLY, e->st);
    launch_sample(e->n, e->S, e->ply & 1, e->ply, temperature, threshold, e->key, e->cfg.game_offset, e->d_actions, e->st);
    e->prof.end(e->st);
}

extern "C" int omok_sample_actions(omok_engine* e, float temperature, int32_t threshold, int32_t* actions) {
    if (!e) return OMOK_ERR_INVALID;
    if (need_reset(e)) return OMOK_ERR_STATE;
    if (!(temperature > 0.0f)) return fail(e, OMOK_ERR_INVALID, "temperature must be > 0");
    HIPCHK(e, hipSetDevice(e->cfg.device));
    enqueue_sample(e, temperature, threshold);
    if (actions) HIPCHK(e, hipMemcpyAsync(actions, e->d_actions, sizeof(int32_t) * e->cfg.games, hipMemcpyDeviceToHost, e->st));
    uint32_t bits = 0;
    if (read_status(e, &bits, nullptr)) return OMOK_ERR_HIP;
    e->sampled = true;
    return tree_error(e, bits);
}

static void enqueue_mirror_and_advance(omok_engine* e, int alive) {
    const int side = e->ply & 1;
    e->prof.begin(PC_PLY, e->st);
    launch_mirror_scan(e->n, e->S, side, e->st);
    k_add_evals<<<1, 64, 0, e->st>>>(e->S.d_count, e->d_evals);
    e->prof.end(e->st);
    net_forward_requests(e->net, e->S, alive, e->st, &e->prof);
    e->prof.begin(PC_PLY, e->st);
    launch_advance(e->n, e->S, side, e->net.p, e->st);
    net_invalidate_sibling_cache(e->net);
    e->prof.end(e->st);
}

extern "C" int omok_advance(omok_engine* e) {
    if (!e) return OMOK_ERR_INVALID;
    if (need_net(e) || need_reset(e)) return OMOK_ERR_STATE;
    if (!e->sampled) return fail(e, OMOK_ERR_STATE, "omok_sample_actions must precede omok_advance");
    HIPCHK(e, hipSetDevice(e->cfg.device));
    uint32_t bits = 0, before = 0, after = 0;
    if (read_status(e, &bits, &before)) return OMOK_ERR_HIP;
    enqueue_mirror_and_advance(e, (int)before);
    if (read_status(e, &bits, &after)) return OMOK_ERR_HIP;
    e->ply_games += before;
    e->finished += (double)before - (double)after;
    e->ply += 1;
    e->sampled = false;
    return tree_error(e, bits);
}

extern "C" int omok_selfplay_run(omok_engine* e, int32_t count, int32_t batch_size, float epsilon, float alpha,
                                 float temperature, int32_t threshold, int32_t max_plies, double* stats) {
    if (!e) return OMOK_ERR_INVALID;
    if (need_net(e) || need_reset(e)) return OMOK_ERR_STATE;
    if (check_exec_args(e, count, batch_size, epsilon, alpha)) return OMOK_ERR_INVALID;
    if (!(temperature > 0.0f)) return fail(e, OMOK_ERR_INVALID, "temperature must be > 0");
    HIPCHK(e, hipSetDevice(e->cfg.device));
    uint32_t bits = 0, alive = 0;
    if (read_status(e, &bits, &alive)) return OMOK_ERR_HIP;
    int plies = 0;
    while (alive > 0 && (max_plies <= 0 || plies < max_plies)) { // trainer.rs:95
        const int rounds = enqueue_execute(e, count, batch_size, epsilon, alpha, (int)alive);
        enqueue_sample(e, temperature, threshold);
        enqueue_mirror_and_advance(e, (int)alive);
        e->sims += (double)rounds * batch_size * alive;
        e->ply_games += alive;
        e->ply += 1;
        plies += 1;
        uint32_t after = 0;
        if (read_status(e, &bits, &after)) return OMOK_ERR_HIP;
        e->finished += (double)alive - (double)after;
        alive = after;
        if (tree_error(e, bits)) return bits & 1u ? OMOK_ERR_OVERFLOW : OMOK_ERR_ILLEGAL;
    }
    e->sampled = false;
    if (stats) return omok_get_stats(e, stats);
    return OMOK_OK;
}


// Slots mode: the engine's `games` slots are kept full.  Same games, same results as an episode of `total_games` games (every game
// is keyed by its index: RNG streams by cfg.game_offset + index and the game's own ply; trees are independent), but a slot whose game
// is over takes the next index instead of idling until the longest game of the episode ends (the last 40 % of an episode's rounds
// hold < 10 % of its rows).  Games start on even engine plies only (one `side` per ply: a slot may wait one ply), finished games'
// transitions are packed out (k_replay_pack records) before their slot is reused.
extern "C" int omok_selfplay_run_slots(omok_engine* e, int32_t total_games, int32_t count, int32_t batch_size, float epsilon, float alpha,
                                       float temperature, int32_t threshold, void* records_dev, int64_t cap_records, int64_t* game_offsets,
                                       int32_t* game_lengths, int32_t* game_status, int64_t* n_records, double* stats) {
    if (!e) return OMOK_ERR_INVALID;
    if (need_net(e) || need_reset(e)) return OMOK_ERR_STATE;
    if (check_exec_args(e, count, batch_size, epsilon, alpha)) return OMOK_ERR_INVALID;
    if (!(temperature > 0.0f)) return fail(e, OMOK_ERR_INVALID, "temperature must be > 0");
    const int G = e->cfg.games;
    if (total_games < G) return fail(e, OMOK_ERR_INVALID, "total_games (%d) must be >= the engine's game slots (%d)", total_games, G);
    if (!records_dev || cap_records < 1) return fail(e, OMOK_ERR_INVALID, "records buffer required (omok_replay_record_bytes per record)");
    if (e->ply != 0) return fail(e, OMOK_ERR_STATE, "omok_selfplay_run_slots starts from a fresh omok_selfplay_reset (ply %d)", e->ply);
    ENTER(e);
    uint8_t* d_mask = nullptr;
    long long *d_slot_off = nullptr, *d_out = nullptr;
    int32_t *d_next = nullptr, *d_new = nullptr;
    SlotMeta* d_meta = nullptr;
    auto cleanup = [&]() { for (void* p : {(void*)d_mask, (void*)d_slot_off, (void*)d_out, (void*)d_next, (void*)d_new, (void*)d_meta}) if (p) hipFree(p); };
    if (hipMalloc(&d_mask, G) != hipSuccess || hipMalloc(&d_slot_off, sizeof(long long) * G) != hipSuccess || hipMalloc(&d_out, 8) != hipSuccess ||
        hipMalloc(&d_next, 4) != hipSuccess || hipMalloc(&d_new, sizeof(int32_t) * G) != hipSuccess ||
        hipMalloc(&d_meta, sizeof(SlotMeta) * (size_t)total_games) != hipSuccess) {
        cleanup();
        return fail(e, OMOK_ERR_HIP, "slots mode: device allocation failed");
    }
    hipMemsetAsync(d_out, 0, 8, e->st);
    hipMemsetAsync(d_meta, 0xFF, sizeof(SlotMeta) * (size_t)total_games, e->st);
    hipMemcpyAsync(d_next, &G, 4, hipMemcpyHostToDevice, e->st); // the reset started games 0 .. G-1
    uint32_t bits = 0, alive = 0;
    if (read_status(e, &bits, &alive)) { cleanup(); return OMOK_ERR_HIP; }
    int rc = OMOK_OK;
    while (alive > 0) {
        const int rounds = enqueue_execute(e, count, batch_size, epsilon, alpha, (int)alive);
        enqueue_sample(e, temperature, threshold);
        enqueue_mirror_and_advance(e, (int)alive);
        e->sims += (double)rounds * batch_size * alive;
        e->ply_games += alive;
        e->ply += 1;
        e->prof.begin(PC_PLY, e->st);
        launch_harvest(e->n, e->S, d_mask, d_slot_off, d_out, d_meta, (uint8_t*)records_dev, cap_records, e->st);
        if ((e->ply & 1) == 0) launch_refill(e->n, e->S, e->d_root_policy, d_next, total_games, d_new, e->st);
        net_invalidate_sibling_cache(e->net);
        e->prof.end(e->st);
        uint32_t after = 0;
        if (read_status(e, &bits, &after)) { cleanup(); return OMOK_ERR_HIP; }
        alive = after;
        if (tree_error(e, bits)) { rc = bits & 1u ? OMOK_ERR_OVERFLOW : OMOK_ERR_ILLEGAL; break; }
        if (alive == 0 && (e->ply & 1)) { // every slot is waiting for an even ply with games left to start: let the ply pass
            int32_t next = 0;
            hipMemcpy(&next, d_next, 4, hipMemcpyDeviceToHost);
            if (next >= total_games) break;
            e->ply += 1;
            launch_refill(e->n, e->S, e->d_root_policy, d_next, total_games, d_new, e->st);
            net_invalidate_sibling_cache(e->net);
            if (read_status(e, &bits, &alive)) { cleanup(); return OMOK_ERR_HIP; }
        }
    }
    e->sampled = false;
    long long out = 0;
    hipMemcpy(&out, d_out, 8, hipMemcpyDeviceToHost);
    if (n_records) *n_records = out;
    if (rc == OMOK_OK && out > cap_records) rc = fail(e, OMOK_ERR_OVERFLOW, "slots mode: %lld records do not fit the buffer (%lld)", out, (long long)cap_records);
    if (game_offsets || game_lengths || game_status) {
        std::vector<SlotMeta> meta((size_t)total_games);
        hipMemcpy(meta.data(), d_meta, sizeof(SlotMeta) * meta.size(), hipMemcpyDeviceToHost);
        for (int i = 0; i < total_games; ++i) {
            if (game_offsets) game_offsets[i] = meta[i].offset;
            if (game_lengths) game_lengths[i] = meta[i].len;
            if (game_status) game_status[i] = meta[i].status;
        }
    }
    e->finished += (double)total_games;
    cleanup();
    if (rc != OMOK_OK) return rc;
    if (stats) return omok_get_stats(e, stats);
    return OMOK_OK;
}

extern "C" int omok_set_episode(omok_engine* e, uint64_t episode) {
    if (!e) return OMOK_ERR_INVALID;
    e->episode = episode;
    return OMOK_OK;
}

// Agent::compute_policy for every game (agent.rs:43-77)
extern "C" int omok_compute_policy(omok_engine* e, float* pi, uint8_t* has_policy) {
    if (!e || !pi) return OMOK_ERR_INVALID;
    if (need_reset(e)) return OMOK_ERR_STATE;
    ENTER(e);
    launch_policy(e->n, e->S, e->ply & 1, e->d_pi, e->d_has, e->st);
    HIPCHK(e, hipMemcpyAsync(pi, e->d_pi, sizeof(float) * (size_t)e->cfg.games * e->hw, hipMemcpyDeviceToHost, e->st));
    if (has_policy) HIPCHK(e, hipMemcpyAsync(has_policy, e->d_has, (size_t)e->cfg.games, hipMemcpyDeviceToHost, e->st));
    return sync_and_check(e, "compute_policy") ? OMOK_ERR_HIP : OMOK_OK;
}

// Externally chosen moves: Agent::ensure_action_exists + Agent::play_action on BOTH agents of every live game
// (agent.rs:144-232; the flow of gui/src/agent.rs:49-66 and benchmark/src/agent.rs:34-50 for a move the search did not pick).
static int stage_actions(omok_engine* e, const int32_t* actions) {
    uint32_t flags[4] = {0, 0, 0, 0};
    HIPCHK(e, hipMemsetAsync(e->d_flags, 0, 16, e->st));
    HIPCHK(e, hipMemcpyAsync(e->d_actions, actions, sizeof(int32_t) * e->cfg.games, hipMemcpyHostToDevice, e->st));
    launch_set_actions(e->n, e->S, e->ply & 1, e->d_actions, e->d_flags, e->st);
    HIPCHK(e, hipMemcpyAsync(flags, e->d_flags, 16, hipMemcpyDeviceToHost, e->st));
    if (sync_and_check(e, "play_actions")) return OMOK_ERR_HIP;
    if (flags[0] || flags[1]) { // nothing has been played yet: drop the staged moves, the position is unchanged
        launch_clear_actions(e->S, e->st);
        if (sync_and_check(e, "play_actions")) return OMOK_ERR_HIP;
        if (flags[0]) return fail(e, OMOK_ERR_ILLEGAL, "%u illegal move(s): cell occupied or out of range (place_stone -> None)", flags[0]);
        return fail(e, OMOK_ERR_INVALID, "%u live game(s) without a move: every live game moves in a ply (all games share the side to move)", flags[1]);
    }
    return OMOK_OK;
}

extern "C" int omok_set_actions(omok_engine* e, const int32_t* actions) {
    if (!e || !actions) return OMOK_ERR_INVALID;
    if (need_reset(e)) return OMOK_ERR_STATE;
    ENTER(e);
    const int rc = stage_actions(e, actions);
    if (rc == OMOK_OK) e->sampled = true;
    return rc;
}

extern "C" int omok_play_actions(omok_engine* e, const int32_t* actions) {
    if (!e || !actions) return OMOK_ERR_INVALID;
    if (need_net(e) || need_reset(e)) return OMOK_ERR_STATE;
    ENTER(e);
    const int rc = stage_actions(e, actions);
    if (rc != OMOK_OK) return rc;
    e->sampled = true;
    return omok_advance(e);
}

// children of a root in insertion order (Node::children of MCTS::root, mcts/src/node.rs:10-21): action, n, w, p
extern "C" int omok_root_children(omok_engine* e, int32_t game, int32_t side, int32_t* actions, uint32_t* n, float* w, float* p, int32_t cap) {
    if (!e || game < 0 || game >= e->cfg.games || (side != 0 && side != 1) || cap < 0) return OMOK_ERR_INVALID;
    ENTER(e);
    const size_t t = (size_t)side * e->cfg.games + game, rp = (size_t)e->rowp, nw2 = 2 * (size_t)e->nw;
    const size_t tn = t * (size_t)e->S.stride_nodes, tt = t * (size_t)e->S.stride_tables;
    NodeHdr h0;
    HIPCHK(e, hipMemcpyAsync(&h0, e->S.hdr + tn, sizeof(h0), hipMemcpyDeviceToHost, e->st));
    if (sync_and_check(e, "root_children")) return OMOK_ERR_HIP;
    if (h0.table == NONE16 || h0.nch == 0) return 0;
    std::vector<uint32_t> cn(rp);
    std::vector<float> cw(rp), pol(rp);
    std::vector<uint8_t> co(rp);
    std::vector<uint64_t> bb(nw2);
    const size_t row = (tt + h0.table) * rp;
    hipMemcpyAsync(cn.data(), e->S.tcn + row, 4 * rp, hipMemcpyDeviceToHost, e->st);
    hipMemcpyAsync(cw.data(), e->S.tcw + row, 4 * rp, hipMemcpyDeviceToHost, e->st);
    hipMemcpyAsync(co.data(), e->S.tcorder + row, rp, hipMemcpyDeviceToHost, e->st);
    hipMemcpyAsync(pol.data(), e->S.policy + tn * rp, 4 * rp, hipMemcpyDeviceToHost, e->st);
    hipMemcpyAsync(bb.data(), e->S.board + tn * nw2, 8 * nw2, hipMemcpyDeviceToHost, e->st);
    if (sync_and_check(e, "root_children")) return OMOK_ERR_HIP;
    for (int a = 0; a < e->hw; ++a) {
        const int rank = co[a];
        if (rank == NONE8 || rank >= cap) continue;
        if (actions) actions[rank] = a;
        if (n) n[rank] = cn[a];
        if (w) w[rank] = cw[a];
        if (p) { // child.p == root.policy[action] (node.rs:76, pme.rs:71-75,256-261)
            const bool occ = ((bb[a / 64] | bb[e->nw + a / 64]) >> (a % 64)) & 1ULL;
            p[rank] = h0.has_policy ? pol[a] : ((occ || h0.legal == 0) ? 0.0f : 1.0f / (float)h0.legal);
        }
    }
    return h0.nch;
}

// Environment::place_stone on caller-held environments (environment/src/lib.rs:104-166), batched
extern "C" int omok_env_place_stone(omok_engine* e, uint8_t* boards, uint8_t* turns, uint16_t* legal, const int32_t* actions,
                                    int32_t batch, int32_t* status_out) {
    if (!e || !boards || !turns || !legal || !actions || !status_out || batch < 1) return OMOK_ERR_INVALID;
    ENTER(e);
    uint8_t *d_boards = nullptr, *d_turns = nullptr;
    uint16_t* d_legal = nullptr;
    int32_t *d_act = nullptr, *d_status = nullptr;
    const size_t B = (size_t)batch;
    HIPCHK(e, hipMalloc((void**)&d_boards, B * e->hw));
    HIPCHK(e, hipMalloc((void**)&d_turns, B));
    HIPCHK(e, hipMalloc((void**)&d_legal, B * 2));
    HIPCHK(e, hipMalloc((void**)&d_act, B * 4));
    HIPCHK(e, hipMalloc((void**)&d_status, B * 4));
    hipMemcpyAsync(d_boards, boards, B * e->hw, hipMemcpyHostToDevice, e->st);
    hipMemcpyAsync(d_turns, turns, B, hipMemcpyHostToDevice, e->st);
    hipMemcpyAsync(d_legal, legal, B * 2, hipMemcpyHostToDevice, e->st);
    hipMemcpyAsync(d_act, actions, B * 4, hipMemcpyHostToDevice, e->st);
    launch_env_place(e->n, d_boards, d_turns, d_legal, d_act, batch, d_status, e->st);
    hipMemcpyAsync(boards, d_boards, B * e->hw, hipMemcpyDeviceToHost, e->st);
    hipMemcpyAsync(turns, d_turns, B, hipMemcpyDeviceToHost, e->st);
    hipMemcpyAsync(legal, d_legal, B * 2, hipMemcpyDeviceToHost, e->st);
    hipMemcpyAsync(status_out, d_status, B * 4, hipMemcpyDeviceToHost, e->st);
    const int rc = sync_and_check(e, "env_place_stone");
    hipFree(d_boards); hipFree(d_turns); hipFree(d_legal); hipFree(d_act); hipFree(d_status);
    return rc ? OMOK_ERR_HIP : OMOK_OK;
}

// ---- step-wise API (parity tests) --------------------------------------------------------------
extern "C" int omok_round_generate(omok_engine* e, int32_t round, int32_t batch_size, float epsilon, float alpha, int32_t* n_requests) {
    if (!e) return OMOK_ERR_INVALID;
    if (need_reset(e)) return OMOK_ERR_STATE;
    if (check_exec_args(e, 1, batch_size, epsilon, alpha)) return OMOK_ERR_INVALID;
    HIPCHK(e, hipSetDevice(e->cfg.device));
    enqueue_round(e, round, batch_size, epsilon, alpha, false, e->cfg.games);
    int32_t cnt = 0;
    HIPCHK(e, hipMemcpyAsync(&cnt, e->S.d_count, sizeof(int32_t), hipMemcpyDeviceToHost, e->st));
    uint32_t bits = 0, alive = 0;
    if (read_status(e, &bits, &alive)) return OMOK_ERR_HIP;
    e->sims += (double)batch_size * alive;
    e->round_reqs = cnt;
    e->round_cap = (int)alive * batch_size;
    if (n_requests) *n_requests = cnt;
    return tree_error(e, bits);
}

static int requests_to_inputs(omok_engine* e, int cnt, float* inputs) {
    if (cnt <= 0) return OMOK_OK;
    launch_encode_requests(e->n, e->S, e->net.in_f32, cnt, e->st);
    HIPCHK(e, hipMemcpyAsync(inputs, e->net.in_f32, sizeof(float) * 3 * (size_t)e->hw * cnt, hipMemcpyDeviceToHost, e->st));
    return sync_and_check(e, "request inputs") ? OMOK_ERR_HIP : OMOK_OK;
}

static int outputs_to_host(omok_engine* e, int cnt, float* p, float* v) {
    if (cnt <= 0) return OMOK_OK;
    float* d_pack = e->net.in_f32;
    const size_t tot = (size_t)cnt * e->hw;
    k_pack_rows<<<(unsigned)((tot + 255) / 256), 256, 0, e->st>>>(e->net.p, d_pack, e->hw, e->rowp, cnt);
    HIPCHK(e, hipMemcpyAsync(p, d_pack, sizeof(float) * tot, hipMemcpyDeviceToHost, e->st));
    if (v) HIPCHK(e, hipMemcpyAsync(v, e->net.v, sizeof(float) * cnt, hipMemcpyDeviceToHost, e->st));
    return sync_and_check(e, "outputs") ? OMOK_ERR_HIP : OMOK_OK;
}

static int inject_outputs(omok_engine* e, int cnt, const float* p, const float* v) {
    if (cnt <= 0) return OMOK_OK;
    float* d_pack = e->net.in_f32;
    HIPCHK(e, hipMemcpyAsync(d_pack, p, sizeof(float) * (size_t)cnt * e->hw, hipMemcpyHostToDevice, e->st));
    const size_t tot = (size_t)cnt * e->rowp;
    k_unpack_rows<<<(unsigned)((tot + 255) / 256), 256, 0, e->st>>>(d_pack, e->net.p, e->hw, e->rowp, cnt);
    if (v) HIPCHK(e, hipMemcpyAsync(e->net.v, v, sizeof(float) * cnt, hipMemcpyHostToDevice, e->st));
    return sync_and_check(e, "inject") ? OMOK_ERR_HIP : OMOK_OK;
}

extern "C" int omok_round_inputs(omok_engine* e, float* inputs) {
    if (!e || !inputs) return OMOK_ERR_INVALID;
    ENTER(e);
    if (e->round_reqs < 0) return fail(e, OMOK_ERR_STATE, "no generated round");
    return requests_to_inputs(e, e->round_reqs, inputs);
}
extern "C" int omok_round_eval(omok_engine* e) {
    ENTER(e);
    if (need_net(e)) return OMOK_ERR_STATE;
    if (e->round_reqs < 0) return fail(e, OMOK_ERR_STATE, "no generated round");
    if (e->round_reqs == 0) return OMOK_OK;
    net_forward_requests(e->net, e->S, e->round_cap, e->st, &e->prof, e->ply & 1);
    return sync_and_check(e, "round_eval") ? OMOK_ERR_HIP : OMOK_OK;
}
extern "C" int omok_round_outputs(omok_engine* e, float* p, float* v) {
    if (!e || !p) return OMOK_ERR_INVALID;
    ENTER(e);
    if (e->round_reqs < 0) return fail(e, OMOK_ERR_STATE, "no generated round");
    return outputs_to_host(e, e->round_reqs, p, v);
}
// the pre-softmax policy logits and the pre-tanh value of the pending round's evaluation (precision evidence for the path the search rounds take)
extern "C" int omok_round_logits(omok_engine* e, float* logits, float* vpre) {
    if (!e || !logits) return OMOK_ERR_INVALID;
    ENTER(e);
    if (e->round_reqs < 0) return fail(e, OMOK_ERR_STATE, "no generated round");
    if (e->round_reqs == 0) return OMOK_OK;
    if (e->net.mode != OMOK_NET_F32 && !net_logits_cover_batch(e->net, e->round_reqs))
        return fail(e, OMOK_ERR_STATE, "omok_round_logits: the round was evaluated in chunks (OMOK_NET_CHUNK): its logits are not kept");
    if (e->net.mode == OMOK_NET_F32 && e->round_reqs > e->net.chunk)
        return fail(e, OMOK_ERR_STATE, "omok_round_logits: OMOK_NET_F32 keeps the logits of its last chunk of %d rows only", e->net.chunk);
    int stride = 0;
    const float* lg = net_logits(e->net, &stride);
    float* d_pack = e->net.in_f32;
    const size_t tot = (size_t)e->round_reqs * e->hw;
    k_pack_rows<<<(unsigned)((tot + 255) / 256), 256, 0, e->st>>>(lg, d_pack, e->hw, stride, e->round_reqs);
    HIPCHK(e, hipMemcpyAsync(logits, d_pack, sizeof(float) * tot, hipMemcpyDeviceToHost, e->st));
    if (vpre) HIPCHK(e, hipMemcpyAsync(vpre, e->net.vpre, sizeof(float) * e->round_reqs, hipMemcpyDeviceToHost, e->st));
    return sync_and_check(e, "round_logits") ? OMOK_ERR_HIP : OMOK_OK;
}
extern "C" int omok_round_inject(omok_engine* e, const float* p, const float* v) {
    if (!e || !p || !v) return OMOK_ERR_INVALID;
    ENTER(e);
    if (e->round_reqs < 0) return fail(e, OMOK_ERR_STATE, "no generated round");
    return inject_outputs(e, e->round_reqs, p, v);
}
extern "C" int omok_round_scatter(omok_engine* e) {
    ENTER(e);
    if (e->round_reqs < 0) return fail(e, OMOK_ERR_STATE, "no generated round");
    if (e->round_reqs > 0) launch_scatter(e->n, e->S, e->ply & 1, e->net.p, e->net.v, e->round_reqs, e->st);
    e->round_reqs = -1;
    return sync_and_check(e, "round_scatter") ? OMOK_ERR_HIP : OMOK_OK;
}

extern "C" int omok_mirror_generate(omok_engine* e, int32_t* n_requests) {
    ENTER(e);
    if (need_reset(e)) return OMOK_ERR_STATE;
    if (!e->sampled) return fail(e, OMOK_ERR_STATE, "omok_sample_actions must precede the mirror step");
    launch_mirror_scan(e->n, e->S, e->ply & 1, e->st);
    k_add_evals<<<1, 64, 0, e->st>>>(e->S.d_count, e->d_evals);
    int32_t cnt = 0;
    HIPCHK(e, hipMemcpyAsync(&cnt, e->S.d_count, sizeof(int32_t), hipMemcpyDeviceToHost, e->st));
    if (sync_and_check(e, "mirror_generate")) return OMOK_ERR_HIP;
    e->mirror_reqs = cnt;
    if (n_requests) *n_requests = cnt;
    return OMOK_OK;
}
extern "C" int omok_mirror_inputs(omok_engine* e, float* inputs) {
    if (!e || !inputs) return OMOK_ERR_INVALID;
    ENTER(e);
    if (e->mirror_reqs < 0) return fail(e, OMOK_ERR_STATE, "no generated mirror batch");
    return requests_to_inputs(e, e->mirror_reqs, inputs);
}
extern "C" int omok_mirror_eval(omok_engine* e) {
    ENTER(e);
    if (need_net(e)) return OMOK_ERR_STATE;
    if (e->mirror_reqs < 0) return fail(e, OMOK_ERR_STATE, "no generated mirror batch");
    if (e->mirror_reqs > 0) net_forward_requests(e->net, e->S, e->mirror_reqs, e->st, &e->prof);
    return sync_and_check(e, "mirror_eval") ? OMOK_ERR_HIP : OMOK_OK;
}
extern "C" int omok_mirror_outputs(omok_engine* e, float* p) {
    if (!e || !p) return OMOK_ERR_INVALID;
    ENTER(e);
    if (e->mirror_reqs < 0) return fail(e, OMOK_ERR_STATE, "no generated mirror batch");
    return outputs_to_host(e, e->mirror_reqs, p, nullptr);
}
extern "C" int omok_mirror_inject(omok_engine* e, const float* p) {
    if (!e || !p) return OMOK_ERR_INVALID;
    ENTER(e);
    if (e->mirror_reqs < 0) return fail(e, OMOK_ERR_STATE, "no generated mirror batch");
    return inject_outputs(e, e->mirror_reqs, p, nullptr);
}
extern "C" int omok_mirror_apply(omok_engine* e) {
    ENTER(e);
    if (e->mirror_reqs < 0) return fail(e, OMOK_ERR_STATE, "no generated mirror batch");
    uint32_t bits = 0, before = 0, after = 0;
    if (read_status(e, &bits, &before)) return OMOK_ERR_HIP;
    launch_advance(e->n, e->S, e->ply & 1, e->net.p, e->st);
    net_invalidate_sibling_cache(e->net);
    if (read_status(e, &bits, &after)) return OMOK_ERR_HIP;
    e->ply_games += before;
    e->finished += (double)before - (double)after;
    e->ply += 1;
    e->sampled = false;
    e->mirror_reqs = -1;
    return tree_error(e, bits);
}

// ---- inspection ----------------------------------------------------------------------------------
extern "C" int omok_alive_count(omok_engine* e) {
    ENTER(e);
    uint32_t bits = 0, alive = 0;
    if (read_status(e, &bits, &alive)) return OMOK_ERR_HIP;
    return (int)alive;
}
extern "C" int omok_current_ply(omok_engine* e) { return e ? e->ply : OMOK_ERR_INVALID; }

extern "C" int omok_game_info(omok_engine* e, uint8_t* alive, uint8_t* status, int32_t* plies) {
    if (!e) return OMOK_ERR_INVALID;
    ENTER(e);
    std::vector<GameState> gs((size_t)e->cfg.games);
    HIPCHK(e, hipMemcpyAsync(gs.data(), e->S.gs, sizeof(GameState) * gs.size(), hipMemcpyDeviceToHost, e->st));
    if (sync_and_check(e, "game_info")) return OMOK_ERR_HIP;
    for (size_t g = 0; g < gs.size(); ++g) {
        if (alive) alive[g] = gs[g].alive;
        if (status) status[g] = gs[g].status;
        if (plies) plies[g] = gs[g].plies;
    }
    return OMOK_OK;
}

extern "C" int omok_tree_root(omok_engine* e, int32_t game, int32_t side, uint32_t* root_n, float* root_w, int32_t* n_nodes, int32_t* n_tables) {
    if (!e || game < 0 || game >= e->cfg.games || (side != 0 && side != 1)) return OMOK_ERR_INVALID;
    ENTER(e);
    TreeState ts;
    HIPCHK(e, hipMemcpyAsync(&ts, e->S.ts + (size_t)side * e->cfg.games + game, sizeof(ts), hipMemcpyDeviceToHost, e->st));
    if (sync_and_check(e, "tree_root")) return OMOK_ERR_HIP;
    if (root_n) *root_n = ts.root_n;
    if (root_w) *root_w = ts.root_w;
    if (n_nodes) *n_nodes = (int32_t)ts.n_nodes;
    if (n_tables) *n_tables = (int32_t)ts.n_tables;
    return OMOK_OK;
}

extern "C" int omok_tree_dump(omok_engine* e, int32_t game, int32_t side, int32_t* ints, float* floats, int32_t cap_nodes) {
    if (!e || !ints || !floats || game < 0 || game >= e->cfg.games || (side != 0 && side != 1)) return OMOK_ERR_INVALID;
    ENTER(e);
    const size_t t = (size_t)side * e->cfg.games + game;
    TreeState ts;
    HIPCHK(e, hipMemcpyAsync(&ts, e->S.ts + t, sizeof(ts), hipMemcpyDeviceToHost, e->st));
    if (sync_and_check(e, "tree_dump")) return OMOK_ERR_HIP;
    const int nn = (int)ts.n_nodes, nt = (int)ts.n_tables;
    if (nn > cap_nodes) return -nn;
    const size_t rp = (size_t)e->rowp, nw2 = 2 * (size_t)e->nw;
    std::vector<NodeHdr> hdr((size_t)nn);
    std::vector<uint64_t> board((size_t)nn * nw2);
    std::vector<float> pol((size_t)nn * rp), cw((size_t)nt * rp);
    std::vector<uint32_t> cn((size_t)nt * rp);
    std::vector<uint8_t> co((size_t)nt * rp);
    const size_t tn = t * (size_t)e->S.stride_nodes, tt = t * (size_t)e->S.stride_tables;
    hipMemcpyAsync(hdr.data(), e->S.hdr + tn, sizeof(NodeHdr) * nn, hipMemcpyDeviceToHost, e->st);
    hipMemcpyAsync(board.data(), e->S.board + tn * nw2, 8 * nw2 * nn, hipMemcpyDeviceToHost, e->st);
    hipMemcpyAsync(pol.data(), e->S.policy + tn * rp, 4 * rp * nn, hipMemcpyDeviceToHost, e->st);
    if (nt) {
        hipMemcpyAsync(cn.data(), e->S.tcn + tt * rp, 4 * rp * nt, hipMemcpyDeviceToHost, e->st);
        hipMemcpyAsync(cw.data(), e->S.tcw + tt * rp, 4 * rp * nt, hipMemcpyDeviceToHost, e->st);
        hipMemcpyAsync(co.data(), e->S.tcorder + tt * rp, rp * nt, hipMemcpyDeviceToHost, e->st);
    }
    if (sync_and_check(e, "tree_dump")) return OMOK_ERR_HIP;
    const int hw = e->hw;
    for (int i = 0; i < nn; ++i) {
        const NodeHdr& h = hdr[i];
        int32_t* o = ints + (size_t)i * 8;
        float* f = floats + (size_t)i * (1 + hw);
        uint32_t n = ts.root_n;
        float w = ts.root_w;
        int order = -1;
        if (i != 0) {
            const size_t slot = (size_t)hdr[h.parent].table * rp + h.action;
            n = cn[slot]; w = cw[slot]; order = co[slot];
        }
        o[0] = h.parent == NONE16 ? -1 : h.parent;
        o[1] = h.action == NONE8 ? -1 : h.action;
        o[2] = h.status; o[3] = h.turn; o[4] = h.legal; o[5] = h.nch; o[6] = (int32_t)n;
        o[7] = (order & 0xffff) | ((int32_t)h.has_policy << 16);
        f[0] = w;
        for (int a = 0; a < hw; ++a) {
            float v;
            if (h.has_policy) v = pol[(size_t)i * rp + a];
            else {
                const bool occ = ((board[(size_t)i * nw2 + a / 64] | board[(size_t)i * nw2 + e->nw + a / 64]) >> (a % 64)) & 1ULL;
                v = (occ || h.legal == 0) ? 0.0f : 1.0f / (float)h.legal;
            }
            f[1 + a] = v;
        }
    }
    return nn;
}

extern "C" int omok_replay_game(omok_engine* e, int32_t game, uint8_t* boards, uint8_t* turns, float* pi, float* z, int32_t cap_plies) {
    if (!e || game < 0 || game >= e->cfg.games) return OMOK_ERR_INVALID;
    ENTER(e);
    GameState gs;
    HIPCHK(e, hipMemcpyAsync(&gs, e->S.gs + game, sizeof(gs), hipMemcpyDeviceToHost, e->st));
    if (sync_and_check(e, "replay_game")) return OMOK_ERR_HIP;
    const int plies = gs.rp_len < e->hw ? gs.rp_len : e->hw;
    const int n = plies < cap_plies ? plies : cap_plies;
    if (n <= 0) return plies;
    const size_t rp = (size_t)e->rowp, nw2 = 2 * (size_t)e->nw, hw = (size_t)e->hw;
    std::vector<uint64_t> b((size_t)n * nw2);
    std::vector<float> ppi((size_t)n * rp);
    const size_t rec = (size_t)game * hw;
    hipMemcpyAsync(b.data(), e->S.rp_board + rec * nw2, 8 * nw2 * n, hipMemcpyDeviceToHost, e->st);
    hipMemcpyAsync(ppi.data(), e->S.rp_pi + rec * rp, 4 * rp * n, hipMemcpyDeviceToHost, e->st);
    if (turns) hipMemcpyAsync(turns, e->S.rp_turn + rec, (size_t)n, hipMemcpyDeviceToHost, e->st);
    if (z) hipMemcpyAsync(z, e->S.rp_z + rec, 4 * (size_t)n, hipMemcpyDeviceToHost, e->st);
    if (sync_and_check(e, "replay_game")) return OMOK_ERR_HIP;
    for (int p = 0; p < n; ++p)
        for (size_t a = 0; a < hw; ++a) {
            if (boards) {
                const bool bl = (b[(size_t)p * nw2 + a / 64] >> (a % 64)) & 1ULL, wh = (b[(size_t)p * nw2 + e->nw + a / 64] >> (a % 64)) & 1ULL;
                boards[(size_t)p * hw + a] = bl ? 1 : (wh ? 2 : 0);
            }
            if (pi) pi[(size_t)p * hw + a] = ppi[(size_t)p * rp + a];
        }
    return plies;
}

extern "C" int32_t omok_replay_record_bytes(const omok_engine* e) {
    if (!e) return OMOK_ERR_INVALID;
    return (e->hw + 1 + 3) / 4 * 4 + 4 * e->hw + 4;
}

extern "C" int64_t omok_replay_pack_dev(omok_engine* e, void* dst_dev, int64_t cap_records) {
    if (!e || !dst_dev || cap_records < 0) return OMOK_ERR_INVALID;
    if (hipSetDevice(e->cfg.device) != hipSuccess) return OMOK_ERR_HIP;
    launch_replay_offsets(e->n, e->S, 1, e->d_aug_offsets, e->st);
    launch_replay_pack(e->n, e->S, e->d_aug_offsets, (uint8_t*)dst_dev, cap_records, e->st);
    long long total = 0;
    if (hipMemcpyAsync(&total, e->d_aug_offsets + e->cfg.games, 8, hipMemcpyDeviceToHost, e->st) != hipSuccess) return OMOK_ERR_HIP;
    if (sync_and_check(e, "replay_pack")) return OMOK_ERR_HIP;
    return total;
}

// ---- replay post-processing (src/trainer.rs:207-324) ----------------------------------------------------------------
extern "C" int64_t omok_replay_augment_dev(omok_engine* e, void* dst_dev, int64_t cap_records) {
    if (!e || !dst_dev || cap_records < 0) return OMOK_ERR_INVALID;
    if (hipSetDevice(e->cfg.device) != hipSuccess) return OMOK_ERR_HIP;
    launch_replay_offsets(e->n, e->S, 6, e->d_aug_offsets, e->st);
    launch_replay_augment(e->n, e->S, e->d_aug_offsets, 0, e->cfg.games, 0, (uint8_t*)dst_dev, cap_records, e->st);
    long long total = 0;
    if (hipMemcpyAsync(&total, e->d_aug_offsets + e->cfg.games, 8, hipMemcpyDeviceToHost, e->st) != hipSuccess) return OMOK_ERR_HIP;
    if (sync_and_check(e, "replay_augment")) return OMOK_ERR_HIP;
    return total;
}

extern "C" int omok_replay_augmented_game(omok_engine* e, int32_t game, uint8_t* boards, uint8_t* turns, float* pi, float* z, int32_t cap_records) {
    if (!e || game < 0 || game >= e->cfg.games || cap_records < 0) return OMOK_ERR_INVALID;
    HIPCHK(e, hipSetDevice(e->cfg.device));
    const size_t hw = (size_t)e->hw, brd = (hw + 1 + 3) / 4 * 4, rec = brd + 4 * hw + 4, max_rec = 6 * hw;
    if (!e->d_aug_scratch && dalloc(e, &e->d_aug_scratch, max_rec * rec)) return OMOK_ERR_HIP;
    long long off[2] = {0, 0};
    launch_replay_offsets(e->n, e->S, 6, e->d_aug_offsets, e->st);
    HIPCHK(e, hipMemcpyAsync(off, e->d_aug_offsets + game, 16, hipMemcpyDeviceToHost, e->st));
    if (sync_and_check(e, "replay_augmented_game")) return OMOK_ERR_HIP;
    const int total = (int)(off[1] - off[0]);
    const int n = total < cap_records ? total : cap_records;
    if (n <= 0) return total;
    launch_replay_augment(e->n, e->S, e->d_aug_offsets, game, 1, off[0], e->d_aug_scratch, (long long)max_rec, e->st);
    std::vector<uint8_t> host((size_t)total * rec);
    HIPCHK(e, hipMemcpyAsync(host.data(), e->d_aug_scratch, host.size(), hipMemcpyDeviceToHost, e->st));
    if (sync_and_check(e, "replay_augmented_game")) return OMOK_ERR_HIP;
    for (int i = 0; i < n; ++i) {
        const uint8_t* r = host.data() + (size_t)i * rec;
        if (boards) memcpy(boards + (size_t)i * hw, r, hw);
        if (turns) turns[i] = r[hw];
        if (pi) memcpy(pi + (size_t)i * hw, r + brd, 4 * hw);
        if (z) memcpy(z + i, r + brd + 4 * hw, 4);
    }
    return total;
}

// fc0 operand rows (the trunk's output as fc0 reads it) of the LAST forward: `rows` rows from `first_row`, omok_operand_row_bytes each
extern "C" int64_t omok_operand_row_bytes(const omok_engine* e) { return e && e->net.mode != OMOK_NET_F32 ? (int64_t)e->net.row_u4 * 16 : OMOK_ERR_INVALID; }
extern "C" int omok_debug_operand_rows(omok_engine* e, int32_t first_row, int32_t rows, void* out) {
    if (!e || !out || first_row < 0 || rows < 0 || first_row + rows > e->net.max_b || e->net.mode == OMOK_NET_F32) return OMOK_ERR_INVALID;
    ENTER(e);
    HIPCHK(e, hipMemcpyAsync(out, (const char*)e->net.a_fc0 + (size_t)first_row * e->net.row_u4 * 16, (size_t)rows * e->net.row_u4 * 16, hipMemcpyDeviceToHost, e->st));
    return sync_and_check(e, "debug_operand_rows") ? OMOK_ERR_HIP : OMOK_OK;
}

extern "C" int omok_debug_set_base_cache(omok_engine* e, int32_t enabled) {
    if (!e) return OMOK_ERR_INVALID;
    e->net.base_cache = enabled != 0;
    net_invalidate_sibling_cache(e->net);
    return OMOK_OK;
}

extern "C" int omok_debug_set_window_rects(omok_engine* e, int32_t enabled) {
    if (!e) return OMOK_ERR_INVALID;
    e->net.win_rects = enabled != 0;
    return OMOK_OK;
}

extern "C" int omok_debug_set_children_kernel(omok_engine* e, int32_t which) {
    if (!e) return OMOK_ERR_INVALID;
    if (which != 1 && which != 2) return fail(e, OMOK_ERR_INVALID, "children kernel %d (1 = k_sib_children, 2 = k_sib_children2)", which);
    if (which == 1 && e->net.mode != OMOK_NET_F32 && e->net.diff_fp6)  // (the mixed format's difference rows are written by k_sib_children2 only: the switch would silently do nothing)
        return fail(e, OMOK_ERR_STATE, "omok_debug_set_children_kernel(1): the engine runs the mixed operand format, whose difference path exists in k_sib_children2 only "
                                       "(create the engine with OMOK_NET_F16X3_FP6 or _F16 to compare the two kernels)");
    e->net.sib_v2 = which == 2;
    net_invalidate_sibling_cache(e->net); // (the two kernels read different base-slot layouts)
    return OMOK_OK;
}

extern "C" int omok_get_stats(omok_engine* e, double* stats) {
    if (!e || !stats) return OMOK_ERR_INVALID;
    HIPCHK(e, hipSetDevice(e->cfg.device));
    if (sync_and_check(e, "get_stats")) return OMOK_ERR_HIP;
    e->prof.resolve();
    unsigned long long ev = 0, by = 0;
    HIPCHK(e, hipMemcpy(&ev, e->d_evals, 8, hipMemcpyDeviceToHost));
    HIPCHK(e, hipMemcpy(&by, e->S.d_bytes, 8, hipMemcpyDeviceToHost));
    for (int i = 0; i < OMOK_STAT_COUNT; ++i) stats[i] = 0.0;
    stats[OMOK_STAT_SIMS] = e->sims;
    stats[OMOK_STAT_EVALS] = e->evals + (double)ev;
    stats[OMOK_STAT_PLY_GAMES] = e->ply_games;
    stats[OMOK_STAT_FINISHED] = e->finished;
    stats[OMOK_STAT_MS_TREE] = e->prof.total_ms(PC_ROUND) + e->prof.total_ms(PC_TREE_OTHER);
    stats[OMOK_STAT_MS_TRUNK] = e->prof.total_ms(PC_TRUNK);
    stats[OMOK_STAT_MS_FC0] = e->prof.total_ms(PC_FC0);
    stats[OMOK_STAT_MS_TAIL] = e->prof.total_ms(PC_TAIL);
    stats[OMOK_STAT_MS_PLY] = e->prof.total_ms(PC_PLY);
    stats[OMOK_STAT_FC0_LAUNCHES] = e->prof.total_launches(PC_FC0);
    stats[OMOK_STAT_FC0_ROWS] = e->evals + (double)ev;
    stats[OMOK_STAT_TREE_BYTES] = (double)by;
    stats[OMOK_STAT_ROUND_LAUNCHES] = e->prof.total_launches(PC_ROUND);
    stats[OMOK_STAT_MS_ROUND] = e->prof.total_ms(PC_ROUND);
    stats[OMOK_STAT_PEAK_NODES] = (double)e->peak_nodes;
    stats[OMOK_STAT_PEAK_TABLES] = (double)e->peak_tables;
    stats[OMOK_STAT_FC0_FORMAT] = e->net.mode == OMOK_NET_F32 ? -1.0 : (e->net.diff_fp6 ? (double)FC0_MIXED : (double)e->net.fc0_fmt);
    stats[OMOK_STAT_PROBE_ROWS] = e->net.probe[6] != 0.0f ? (double)e->net.probe[0] : 0.0;
    stats[OMOK_STAT_PROBE_DP_FP6] = e->net.probe[1];
    stats[OMOK_STAT_PROBE_DV_FP6] = e->net.probe[2];
    stats[OMOK_STAT_PROBE_DP_F16] = e->net.probe[3];
    stats[OMOK_STAT_PROBE_DV_F16] = e->net.probe[4];
    stats[OMOK_STAT_PROBE_LIMIT] = NET_PROBE_LIMIT;
    stats[OMOK_STAT_PROBE_LOGIT_MAX] = e->net.probe[5];
    stats[OMOK_STAT_CHILDREN2_LAUNCHES] = e->net.children_launches[0];
    stats[OMOK_STAT_CHILDREN1_LAUNCHES] = e->net.children_launches[1];
    stats[OMOK_STAT_PROBE_DLOGIT_FP6] = e->net.probe[7];
    stats[OMOK_STAT_PROBE_DLOGIT_F16] = e->net.probe[8];
    stats[OMOK_STAT_PROBE_ROUND_ROWS] = e->net.probe[9];
    for (int i = 0; i < 9; ++i) stats[OMOK_STAT_PROBE_ROUND_FP6 + i] = e->net.probe[10 + i];
    stats[OMOK_STAT_PROBE_LOGIT_LIMIT] = NET_PROBE_LOGIT_LIMIT;
    stats[OMOK_STAT_PROBE_OUTSIDE] = (double)e->net.probe_outside;
    if (e->net.d_work) { // executed-work counters of the sibling rounds (device side: k_group, k_bin_prefix)
        unsigned long long w[NET_WORK_COUNT];
        if (hipMemcpy(w, e->net.d_work, sizeof(w), hipMemcpyDeviceToHost) != hipSuccess) return OMOK_ERR_HIP;
        for (int i = 0; i < 10; ++i) stats[OMOK_STAT_WORK_DIFF_RUNS + i] = (double)w[i];
    }
    return OMOK_OK;
}

extern "C" int omok_reset_stats(omok_engine* e) {
    ENTER(e);
    if (sync_and_check(e, "reset_stats")) return OMOK_ERR_HIP;
    e->prof.resolve();
    for (int i = 0; i < PC_COUNT; ++i) { e->prof.ms[i] = 0; e->prof.launches[i] = 0; e->prof.ms_s[i] = 0; e->prof.launches_s[i] = 0; }
    e->prof.rounds_seen = e->prof.rounds_timed = 0;
    e->sims = e->evals = e->ply_games = e->finished = 0;
    e->peak_nodes = e->peak_tables = 0;
    e->net.children_launches[0] = e->net.children_launches[1] = 0.0;
    hipMemset(e->d_evals, 0, 16);
    hipMemset(e->S.d_bytes, 0, 16);
    if (e->net.d_work) hipMemset(e->net.d_work, 0, sizeof(unsigned long long) * NET_WORK_COUNT);
    return OMOK_OK;
}

extern "C" int omok_set_profiling(omok_engine* e, int32_t enabled) {
    if (!e) return OMOK_ERR_INVALID;
    if (enabled < 0) return OMOK_ERR_INVALID;
    e->prof.resolve();
    e->prof.enabled = enabled != 0;
    e->prof.every = enabled > 1 ? enabled : 1;
    return OMOK_OK;
}
