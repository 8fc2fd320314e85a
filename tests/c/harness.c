/* C caller of the boundary (include/omok_mi355x.h): what a Rust `extern "C"` user does, verified with a C compiler.
 * Plays two plies of two 9x9 games through omok_create -> omok_net_load / commit -> omok_selfplay_reset ->
 * { omok_execute, omok_sample_actions, omok_advance } and checks what a host of src/trainer.rs:95-205 relies on:
 * return codes, the moves land on the replayed boards, the side to move alternates, visit counts add up.
 * Built by __graft_entry__.build() (gcc, linked against ../../omok-ai_amd/libomok_mi355x.so); run by tests/test_c_harness.py.
 * Exit codes: 0 = all checks passed on a GPU; 3 = no HIP device: omok_create failed with OMOK_ERR_HIP and a message (the error
 * path of the ABI, which is all a box without a GPU can exercise); 1 = a check failed. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "omok_mi355x.h"

#define CHECK(cond, ...) do { if (!(cond)) { fprintf(stderr, "harness: FAILED %s:%d: ", __FILE__, __LINE__); fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); return 1; } } while (0)
#define CALL(expr) do { int rc_ = (expr); CHECK(rc_ >= 0, "%s -> %d (%s)", #expr, rc_, omok_last_error(e)); } while (0)

static unsigned long long lcg = 0x9E3779B97F4A7C15ULL;
static float uniform(void) { /* (-1, 1) */
    lcg = lcg * 6364136223846793005ULL + 1442695040888963407ULL;
    return (float)((double)(lcg >> 11) / 9007199254740992.0 * 2.0 - 1.0);
}

int main(void) {
    enum { N = 9, HW = N * N, G = 2, SIMS = 32, K = 8 };
    omok_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.board_size = N; cfg.games = G; cfg.max_nodes = 256; cfg.max_tables = 128; cfg.max_batch_k = K;
    cfg.device = 0; cfg.net_mode = OMOK_NET_F16X3; cfg.seed = 7; cfg.game_offset = 0;
    omok_engine* e = NULL;
    int rc = omok_create(&cfg, &e);
    if (rc == OMOK_ERR_HIP) {
        const char* msg = omok_last_error(NULL);
        CHECK(e == NULL && msg && strlen(msg) > 0, "no engine and a message expected when there is no device");
        printf("harness: no HIP device: omok_create -> OMOK_ERR_HIP (%s)\n", msg);
        return 3;
    }
    CHECK(rc == OMOK_OK && e, "omok_create -> %d (%s)", rc, omok_last_error(NULL));
    CHECK(omok_execute(e, SIMS, K, 0.25f, 0.03f) == OMOK_ERR_STATE, "a search before the net is loaded must be refused (call order)");

    /* 31 variables in Network::variables order; any finite values do for plumbing: He-like scale on the matrices, zero biases */
    const int nt = omok_net_num_tensors();
    CHECK(nt == 31, "omok_net_num_tensors = %d", nt);
    for (int i = 0; i < nt; ++i) {
        const long long len = omok_net_tensor_size(e, i);
        CHECK(len > 0, "tensor %d size %lld", i, len);
        float* w = (float*)malloc(sizeof(float) * (size_t)len);
        const float scale = len > 4096 ? 0.02f : (len > 512 ? 0.15f : 0.0f);
        for (long long j = 0; j < len; ++j) w[j] = scale * uniform();
        CALL(omok_net_load(e, i, w, len));
        CHECK(omok_net_load(e, i, w, len + 1) == OMOK_ERR_INVALID, "a wrong element count must be refused");
        free(w);
    }
    CALL(omok_net_commit(e));

    CALL(omok_selfplay_reset(e));
    CHECK(omok_alive_count(e) == G, "alive %d", omok_alive_count(e));
    int32_t moves[2][G];
    for (int ply = 0; ply < 2; ++ply) {
        CALL(omok_execute(e, SIMS, K, 0.25f, 0.03f));
        for (int g = 0; g < G; ++g) { /* MCTS::root of the side to move: n = simulations so far, children visits add up (agent.rs:43-77) */
            uint32_t root_n = 0, cn[HW];
            float root_w = 0.0f;
            int32_t nn = 0, ntab = 0, acts[HW];
            CALL(omok_tree_root(e, g, ply & 1, &root_n, &root_w, &nn, &ntab));
            const int nch = omok_root_children(e, g, ply & 1, acts, cn, NULL, NULL, HW);
            CHECK(nch > 0 && nch <= HW && nn > 1, "game %d: %d root children, %d nodes", g, nch, nn);
            unsigned long long sum = 0;
            for (int c = 0; c < nch; ++c) sum += cn[c];
            CHECK(sum >= (unsigned long long)SIMS - 1 && sum <= root_n, "game %d ply %d: child visits %llu, root n %u", g, ply, sum, root_n);
        }
        float pi[G * HW];
        uint8_t has[G];
        CALL(omok_compute_policy(e, pi, has));
        for (int g = 0; g < G; ++g) {
            float s = 0.0f;
            for (int a = 0; a < HW; ++a) s += pi[g * HW + a];
            CHECK(has[g] == 1 && s > 0.999f && s < 1.001f, "game %d: compute_policy sums to %f", g, s);
        }
        CALL(omok_sample_actions(e, 1.0f, 30, moves[ply]));
        for (int g = 0; g < G; ++g) CHECK(moves[ply][g] >= 0 && moves[ply][g] < HW && pi[g * HW + moves[ply][g]] > 0.0f, "game %d: move %d", g, moves[ply][g]);
        CALL(omok_advance(e));
        CHECK(omok_current_ply(e) == ply + 1, "ply %d", omok_current_ply(e));
    }
    for (int g = 0; g < G; ++g) { /* Transition{env, policy, z} (trainer.rs:20-24,169-173): env BEFORE the move */
        uint8_t boards[4 * HW], turns[4];
        float pi[4 * HW], z[4];
        const int plies = omok_replay_game(e, g, boards, turns, pi, z, 4);
        CHECK(plies == 2, "game %d: %d transitions", g, plies);
        CHECK(turns[0] == 0 && turns[1] == 1, "game %d: turns %d %d (Black moves first, environment/src/lib.rs:73-79)", g, turns[0], turns[1]);
        int stones0 = 0, stones1 = 0;
        for (int a = 0; a < HW; ++a) { stones0 += boards[a] != 0; stones1 += boards[HW + a] != 0; }
        CHECK(stones0 == 0 && stones1 == 1 && boards[HW + moves[0][g]] == 1, "game %d: the first move must be a Black stone on the second transition's board", g);
        CHECK(moves[1][g] != moves[0][g], "game %d: the second move repeats the first", g);
    }
    { /* an occupied cell is Option::None: OMOK_ERR_ILLEGAL, nothing changes */
        int32_t bad[G];
        for (int g = 0; g < G; ++g) bad[g] = moves[0][g];
        CHECK(omok_play_actions(e, bad) == OMOK_ERR_ILLEGAL && omok_current_ply(e) == 2, "an occupied cell must be refused and leave the games unchanged");
    }
    double stats[OMOK_STAT_COUNT];
    CALL(omok_get_stats(e, stats));
    CHECK(stats[OMOK_STAT_SIMS] == 2.0 * G * SIMS, "simulations %f", stats[OMOK_STAT_SIMS]);
    omok_destroy(e);
    printf("harness OK: 2 plies x %d games through the C ABI\n", G);
    return 0;
}
