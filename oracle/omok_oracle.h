/*
 * omok_oracle.h — CPU ORACLE (TEST INFRASTRUCTURE ONLY, NOT PRODUCT CODE).
 *
 * Plain-C restatement of the self-play hot path of AcrylicShrimp/omok-ai:
 *   environment/src/lib.rs:62-194            (board, place_stone, exact-five check, encode_board)
 *   mcts/src/lib.rs:13-93, node.rs:10-100    (tree, select_leaf, expand, propagate, transition)
 *   alpha-zero/src/parallel_mcts_executor.rs:26-286 (round-based executor, PUCT, noise, scatter)
 *   alpha-zero/src/agent.rs:16-232           (Agent: compute_policy, sample_action, ensure/play)
 *   alpha-zero/src/encoder.rs:10-46          (NN input layout)
 *   alpha-zero/src/network.rs:51-262 + network-utils/src/lib.rs (fp32 policy/value net forward)
 *   src/trainer.rs:81-205                    (self-play ply loop, two trees per game)
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The product (omok-ai_amd/csrc) never links, includes or calls anything in oracle/.
 *
 * PINNING: the reference cannot be built here (no rustc/cargo/libtensorflow).  The rules part
 * is pinned by the reference's own 8 environment tests (environment/src/lib.rs:201-426) and
 * the 5 symmetry tests (src/utils.rs:70-108), transliterated in tests/.  The reference has NO
 * tests for mcts / executor / agent / network and its RNG is unseeded (thread_rng), so for
 * those parts parity is UNPINNED by the reference: they are pinned by this restatement
 * (each function cites the file:line it follows) and by tests/golden fixtures; the fp32 net is
 * cross-checked against an independent torch implementation (tools/make_golden.py).
 *
 * RNG: the reference has no seeds.  This build defines the stream: Philox4x32-10 keyed by the
 * 64-bit seed, counter = (index, ply, tree_global, purpose).  See orc_philox / DESIGN.md.
 */
#ifndef OMOK_ORACLE_H
#define OMOK_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_N 15
#define ORC_MAX_HW 225
#define ORC_EPS 1.1920928955078125e-7f /* f32::EPSILON */

/* enum ABI: declaration order of environment/src/lib.rs:5-9,22-25,46-51 */
enum { ORC_EMPTY = 0, ORC_BLACK = 1, ORC_WHITE = 2 };
enum { ORC_TURN_BLACK = 0, ORC_TURN_WHITE = 1 };
enum { ORC_IN_PROGRESS = 0, ORC_DRAW = 1, ORC_BLACK_WIN = 2, ORC_WHITE_WIN = 3 };
enum { ORC_MODE_PLAYER = 0, ORC_MODE_OPPONENT = 1 };

typedef struct {
    int32_t n;      /* board side */
    uint8_t turn;   /* side to move */
    uint16_t legal; /* legal_move_count */
    uint8_t board[ORC_MAX_HW];
} orc_env;

/* ---- environment (environment/src/lib.rs) ---- */
void orc_env_init(orc_env* e, int n);
int orc_env_place_stone(orc_env* e, int index);                 /* status, or -1 for None */
void orc_env_encode_board(const orc_env* e, int turn, float* dst /*2*HW*/);
void orc_encode_nn_input(const orc_env* e, int mode, float* dst /*3*HW*/);

/* ---- symmetry helpers (src/utils.rs:1-64), "next" row, pinned by its 5 tests ---- */
void orc_rotate_90(const float* src, float* dst, int size);
void orc_rotate_180(const float* src, float* dst, int size);
void orc_rotate_270(const float* src, float* dst, int size);
void orc_flip_horizontal(const float* src, float* dst, int size);
void orc_flip_vertical(const float* src, float* dst, int size);
/* replay post-processing of one game (src/trainer.rs:207-324): z back-fill + 5 augmentations per transition */
void orc_replay_postprocess(int n, int len, const uint8_t* boards, const uint8_t* turns, const float* pi, const float* z_in,
                            uint8_t* boards_out, uint8_t* turns_out, float* pi_out, float* z_out);

/* ---- RNG contract ---- */
enum { ORC_RNG_EXPAND = 1, ORC_RNG_NOISE = 2, ORC_RNG_SAMPLE = 3 };
uint64_t orc_stream_key(uint64_t seed, uint64_t episode); /* seed + episode * 0x9E3779B97F4A7C15: the Philox key of an episode */
void orc_philox(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t out[4]);
double orc_det_log(double x);
double orc_det_exp(double x);
float orc_det_expf(float x);
float orc_gamma(float alpha, uint64_t seed, uint32_t cell, uint32_t ply, uint32_t tree_global);

/* ---- net (alpha-zero/src/network.rs) ---- */
typedef struct orc_net orc_net;
orc_net* orc_net_create(int n);
void orc_net_destroy(orc_net* net);
int orc_net_num_tensors(void);                       /* 31 */
int64_t orc_net_tensor_size(const orc_net* net, int idx);
int orc_net_load(orc_net* net, int idx, const float* data, int64_t count);
/* in: [B][3*HW] (encoder.rs layout), p: [B][HW] softmax probabilities, v: [B] tanh */
void orc_net_forward(const orc_net* net, const float* in, int B, float* p, float* v, int threads);
/* the same forward, also returning logits [B][hw] (in front of the softmax) and vpre [B] (in front of tanh) */
void orc_net_forward_logits(const orc_net* net, const float* in, int B, float* p, float* v, float* logits, float* vpre, int threads);

/* ---- self-play state: G games x two trees (src/trainer.rs:81-94) ---- */
typedef struct orc_sp orc_sp;
orc_sp* orc_sp_create(int n, int games, int cap_nodes, int cap_tables, uint64_t seed, int64_t game_offset);
void orc_sp_destroy(orc_sp* sp);
/* Agent::new for every tree: root policy = raw evaluate_p on the empty board (agent.rs:16-35).  Every reset is one
 * trainer iteration (src/trainer.rs:74-93): it takes the episode counter as its RNG stream and advances it (the first
 * reset after create is episode 0); orc_sp_set_episode sets the index the NEXT reset will use (resume). */
void orc_sp_reset(orc_sp* sp, const float* root_policy /*HW*/);
void orc_sp_set_episode(orc_sp* sp, uint64_t episode);
/* threads > 1: round_generate / round_scatter loop over the games under OpenMP (the reference's rayon par_iter, pme.rs:200-205); same results as threads = 1 */
void orc_sp_set_threads(orc_sp* sp, int threads);
int orc_sp_ply(const orc_sp* sp);
int orc_sp_alive_count(const orc_sp* sp);
int orc_sp_game_alive(const orc_sp* sp, int game);
int orc_sp_game_status(const orc_sp* sp, int game);
int orc_sp_game_plies(const orc_sp* sp, int game);
int orc_sp_error(const orc_sp* sp);
/* One round of ParallelMCTSExecutor::execute (pme.rs:44-213) on the side-to-move trees:
 * round 0 applies Dirichlet noise first.  Writes the encoded NN inputs of the requests
 * (tree order, then sim order) to `inputs` ([max_req][3*HW]) and returns the request count. */
int orc_sp_round_generate(orc_sp* sp, int round, int batch_size, float epsilon, float alpha,
                          float* inputs, int max_req);
/* scatter of the same round (pme.rs:222-265): p [B][HW], v [B] in request order */
void orc_sp_round_scatter(orc_sp* sp, const float* p, const float* v);
/* request r of the last generated round: tree (game) index and node index */
void orc_sp_request_info(const orc_sp* sp, int r, int* game, int* node);
/* trainer.rs:131-161: sample_action for every alive game, records the transition.
 * actions[g] = -1 for finished games. */
void orc_sp_sample(orc_sp* sp, float temperature, int threshold, int32_t* actions /*G*/);
/* trainer.rs:163-166 part 1: NN inputs (Opponent mode) for ensure_action_exists of every
 * alive game, in game order; returns the count */
int orc_sp_mirror_generate(orc_sp* sp, float* inputs, int max_req);
/* play_action on own tree, ensure_action_exists + play_action on the opponent tree,
 * finished games retire (trainer.rs:156-201). p: [count][HW] from evaluate_p */
void orc_sp_advance(orc_sp* sp, const float* p);

/* externally chosen moves (gui/src/agent.rs:49-66 style callers): actions[g] for every alive game replaces orc_sp_sample;
 * the following orc_sp_mirror_generate / orc_sp_advance then run ensure_action_exists + play_action on BOTH agents of the
 * game (agent.rs:144-232) and record no transition. */
void orc_sp_set_actions(orc_sp* sp, const int32_t* actions /*G*/);
/* Agent::compute_policy (agent.rs:43-77) of the side-to-move agent: returns 0 for None */
int orc_sp_compute_policy(const orc_sp* sp, int game, float* policy /*HW*/);

/* canonical tree dump; side 0 = black agent's tree, 1 = white agent's tree.
 * ints: [n_nodes][8] = parent, action, status, turn, legal, nch, n, order | has_policy<<16
 * floats: [n_nodes][1+HW] = w, effective policy row.  returns n_nodes (or -needed if cap too small) */
int orc_sp_tree_dump(const orc_sp* sp, int game, int side, int32_t* ints, float* floats, int cap_nodes);
void orc_sp_tree_root(const orc_sp* sp, int game, int side, uint32_t* root_n, float* root_w, int* n_nodes, int* n_tables);
/* replay (s, pi, z) of a game: boards [plies][HW] u8, turns [plies], pi [plies][HW], z [plies]
 * (z as recorded at play time, trainer.rs:156-173; no back-fill). returns plies */
int orc_sp_replay(const orc_sp* sp, int game, uint8_t* boards, uint8_t* turns, float* pi, float* z, int cap_plies);

/* Whole self-play episode with the oracle net as evaluator (CPU baseline, bench.py only).
 * Runs at most max_plies plies (<=0: to the end).  stats[0]=sims, [1]=nn evals, [2]=plies*games,
 * [3]=finished games, [4]=seconds in net, [5]=seconds total */
int orc_selfplay_run(orc_sp* sp, const orc_net* net, int count, int batch_size, float epsilon,
                     float alpha, float temperature, int threshold, int max_plies, int threads,
                     double* stats);

/* ---- oracle No. 2 (literal.c): the same path restated with the reference's own data structures (pointer nodes that
 *      store p, explicit placeholder policies and refresh loops, recursive free, the trainer's swap_remove vectors).
 *      Requests / rows are in the reference's SLOT order; *_games report the game id of each row. ---- */
typedef struct lit_sp lit_sp;
lit_sp* lit_create(int n, int games, uint64_t seed, int64_t game_offset);
void lit_destroy(lit_sp* sp);
void lit_set_episode(lit_sp* sp, uint64_t episode);
void lit_reset(lit_sp* sp, const float* root_policy);
int lit_ply(const lit_sp* sp);
int lit_error(const lit_sp* sp);
int lit_alive_count(const lit_sp* sp);
int lit_game_alive(const lit_sp* sp, int game);
int lit_game_status(const lit_sp* sp, int game);
int lit_game_plies(const lit_sp* sp, int game);
long lit_live_nodes(const lit_sp* sp);
int lit_round_generate(lit_sp* sp, int round, int batch_size, float epsilon, float alpha, float* inputs, int32_t* req_games, int max_req);
void lit_round_scatter(lit_sp* sp, const float* p, const float* v);
/* MCTSExecutor::run (alpha-zero/src/mcts_executor.rs:29-255) on game 0's side-to-move agent under a GIVEN interleaving of its tasks'
 * simulations: order[i] = task w (round group * waves + w) whose next simulation runs i-th; requests come back in task order then
 * simulation order; the scatter writes all policies, then propagates in the order of its own list.  See oracle/literal.c. */
void lit_shared_noise(lit_sp* sp, float epsilon, float alpha);
int lit_shared_group_generate(lit_sp* sp, int group, int waves, int rounds_total, int batch_size, const uint8_t* order, int n_order,
                              float* inputs /*[max_req][3 HW] or NULL*/, int max_req);
void lit_shared_group_scatter(lit_sp* sp, const float* p, const float* v, const uint8_t* order, int n_order);
void lit_sample(lit_sp* sp, float temperature, int threshold, int32_t* actions /*G, by game id*/);
void lit_set_actions(lit_sp* sp, const int32_t* actions /*G, by game id*/);
int lit_mirror_generate(lit_sp* sp, float* inputs, int32_t* row_games, int max_req);
void lit_advance(lit_sp* sp, const float* p, int external);
int lit_compute_policy(const lit_sp* sp, int game, float* policy);
int lit_tree_dump(const lit_sp* sp, int game, int side, int32_t* ints, float* floats, int cap_nodes);
int lit_tree_priors(const lit_sp* sp, int game, int side, float* p_out, float* parent_policy_at_action, int cap_nodes);
int lit_replay(const lit_sp* sp, int game, uint8_t* boards, uint8_t* turns, float* pi, float* z, int cap_plies);

#ifdef __cplusplus
}
#endif
#endif
