/*
 * omok_mi355x.h — C ABI of the MI355X-native self-play engine (libomok_mi355x.so).
 *
 * Drop-in boundary for the `environment` + `mcts` + `alpha-zero` self-play path of
 * AcrylicShrimp/omok-ai.  The reference has no FFI; the path sits behind Rust crate `pub` APIs.
 * Each entry point below names the reference interface it replaces (file:line in the reference
 * tree).  Plain pointers and sizes only; every buffer is caller-owned host memory unless the
 * name ends in `_dev`.  All device state is owned by the opaque handle.
 *
 * Conventions
 *   - return value: 0 = OK, < 0 = error (OMOK_ERR_*); omok_last_error() gives the text.
 *     Nothing throws or aborts across this boundary (the reference returns Result<_, Status>
 *     / Option and its callers unwrap()).
 *   - enums are the reference's declaration order (environment/src/lib.rs:5-9,22-25,46-51):
 *       Stone {Empty=0, Black=1, White=2}; Turn {Black=0, White=1};
 *       GameStatus {InProgress=0, Draw=1, BlackWin=2, WhiteWin=3}; Option::None -> -1.
 *   - a handle is single-owner (one host thread per GPU); calls block until the work they
 *     describe is complete (mirrors Session::run) unless documented otherwise.
 *   - RNG: the reference is unseeded (thread_rng).  This library defines the stream
 *     (Philox4x32-10, see DESIGN.md "RNG contract"); `seed` and `game_offset` select it.
 */
#ifndef OMOK_MI355X_H
#define OMOK_MI355X_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OMOK_OK 0
#define OMOK_ERR_INVALID (-1)     /* bad argument */
#define OMOK_ERR_HIP (-2)         /* HIP runtime failure (no GPU, OOM, launch error) */
#define OMOK_ERR_STATE (-3)       /* call order violated / net not loaded */
#define OMOK_ERR_OVERFLOW (-4)    /* a tree arena (max_nodes / max_tables) overflowed */
#define OMOK_ERR_ILLEGAL (-5)     /* illegal game operation (Option::None in the reference) */

#define OMOK_MODE_PLAYER 0   /* EnvTurnMode::Player   (alpha-zero/src/encoder.rs:4-8) */
#define OMOK_MODE_OPPONENT 1 /* EnvTurnMode::Opponent */

#define OMOK_NET_F16X3 0 /* split-operand MFMA (x = hi + lo, hi = f16(x): f16 main term + two correction terms, fp32 accumulate).  The
                            correction terms of trunk, fc1 and heads are f16; those of fc0 (68 % of the flops) are block-scaled fp6 (products good to
                            ~2^-15) or f16 (~2^-22, ~2x the fc0 time): omok_net_commit evaluates a fixed probe set of 2048 positions in both
                            formats and with the fp32 kernels and keeps fp6 only while its worst |dp|, |dv| stay within 3e-4 = 0.3 of the
                            1e-3 contract on AgentModel::evaluate_pv's outputs (OMOK_STAT_FC0_FORMAT / OMOK_STAT_PROBE_*) */
#define OMOK_NET_F32 1   /* plain fp32 VALU kernels (debug / A-B reference on the GPU) */
#define OMOK_NET_F16X3_ROWS 2 /* OMOK_NET_F16X3 with every request row evaluated on its own: at board_size 15 the search rounds of
                                 OMOK_NET_F16X3 evaluate sibling requests as one base position + per-child differences, so a
                                 row's p / v carry rounding that depends on its siblings (~5e-5, inside the 1e-3 contract); _ROWS switches
                                 that off (results bit-identical to omok_evaluate_pv of the same position), at ~1.7x the net time */

#define OMOK_NET_F16X3_FP6 3 /* OMOK_NET_F16X3 with fc0's correction terms forced to block-scaled fp6 (no probe) */
#define OMOK_NET_F16X3_F16 4 /* ... forced to f16 */
#define OMOK_NET_F16X3_MIXED 5 /* ... forced to the mixed format: full operand rows (base positions of sibling rounds, single rows, omok_evaluate_pv) with f16
                                  correction terms, the 7x7-window DIFFERENCE rows of sibling rounds -- where the time goes -- with block-scaled fp6 ones: the
                                  quantisation error then scales with the differences, not with the activations */

#define OMOK_MAX_ARENA 16384 /* largest max_nodes / max_tables: node and table indices are 16-bit, and the re-rooting kernel keeps
                               3 B per node + 2 B per table of scratch in LDS (82 KiB at the maximum, inside gfx950's 160 KiB) */

typedef struct omok_engine omok_engine;

typedef struct {
    int32_t board_size;  /* N: 9 (reference, environment/src/lib.rs:70) or 15 */
    int32_t games;       /* G concurrent games = episode_count (src/config.rs:90); two trees each */
    int32_t max_nodes;   /* per-tree node arena, 2 .. OMOK_MAX_ARENA (omok_create rejects more, and sizes whose re-rooting scratch
                            does not fit the device's LDS) */
    int32_t max_tables;  /* per-tree child-table arena, 1 .. OMOK_MAX_ARENA */
    int32_t max_batch_k; /* largest evaluate_batch_size that will be used (<= 64) */
    int32_t device;      /* HIP device ordinal */
    int32_t net_mode;    /* OMOK_NET_* */
    int32_t max_tree_waves; /* 0, or the largest `waves` omok_execute_shared will be called with (<= 16): sizes the net batch */
    uint64_t seed;       /* RNG seed; the Philox key of episode i is seed + i * 0x9E3779B97F4A7C15 (omok_set_episode) */
    int64_t game_offset; /* global id of game 0 (multi-GPU sharding: rank * games) */
} omok_config;

/* ---- lifetime ------------------------------------------------------------------------- */
int omok_create(const omok_config* cfg, omok_engine** out);
void omok_destroy(omok_engine* e);
const char* omok_last_error(const omok_engine* e); /* e may be NULL: error of the last failed create */

/* ---- policy/value net: AgentModel (alpha-zero/src/agent_model.rs:105-134) over Network
 *      (alpha-zero/src/network.rs:51-262).  31 tensors in the reference's variable order
 *      (network.rs:78-79,113-122,149-150,162-163,201-202,240-241), conv kernels HWIO, fc [in,out].
 *      This is also the positional order of ModelIO::load (alpha-zero/src/model_io.rs:92-120). */
int omok_net_num_tensors(void);
int64_t omok_net_tensor_size(const omok_engine* e, int index);
int omok_net_load(omok_engine* e, int index, const float* data, int64_t count);
int omok_net_commit(omok_engine* e); /* pack into MFMA operand layouts; required before any eval */
/* ModelIO::load (alpha-zero/src/model_io.rs:92-120): reads the reference's weights file = bincode 1.3.3 default
 * encoding (little-endian, fixed-width u64 lengths) of SavedData{variable_names: Vec<String>, parameters: Vec<Vec<f32>>}
 * (model_io.rs:20-24).  Loading is POSITIONAL like the reference's zip over `parameters` (:98): names are ignored,
 * parameters beyond the 31st are ignored, fewer than 31 or a length that differs from the variable's element count is an
 * error (the reference fails in session.run / copy_from_slice).  Commits the net on success. */
int omok_net_load_file(omok_engine* e, const char* path);
/* ModelIO::save (model_io.rs:59-90): writes the same format from the tensors currently loaded (canonical names
 * conv_w, conv_b, residual_{i}_..., fc0_w, ...; the reference stores TF-uniquified names and never reads them back). */
int omok_net_save_file(omok_engine* e, const char* path);
/* AgentModel::evaluate_pv (agent_model.rs:116-134): in [B][N][N][3] f32 (encoder.rs layout),
 * p [B][N*N] softmax probabilities, v [B] tanh.  evaluate_p (:105-114) = same with v NULL. */
int omok_evaluate_pv(omok_engine* e, const float* in, int32_t batch, float* p, float* v);
/* The same forward, returning what sits in front of the last two ops of the graph: logits [B][N*N] = input of the Softmax
 * (network.rs:236-247), vpre [B] = input of the Tanh (network.rs:197-200; may be NULL).  Precision evidence / debugging: the
 * reference API has no such call. */
int omok_evaluate_logits(omok_engine* e, const float* in, int32_t batch, float* logits, float* vpre);

/* ---- environment crate on device (environment/src/lib.rs:62-166), batched.
 *      Plays `len` moves per row from Environment::new(); status_out[b][i] is the
 *      Option<GameStatus> of move i (-1 = None: occupied cell, the board is left unchanged).
 *      boards_out [B][N*N] Stone bytes, turns_out [B], legal_out [B] (legal_move_count). */
int omok_env_play(omok_engine* e, const int32_t* moves, int32_t batch, int32_t len,
                  int32_t* status_out, uint8_t* boards_out, uint8_t* turns_out, uint16_t* legal_out);
/* encode_nn_input (alpha-zero/src/encoder.rs:10-46) for `batch` environments given as
 * Stone-byte boards + side to move; out [batch][N][N][3] f32. */
int omok_encode_nn_input(omok_engine* e, const uint8_t* boards, const uint8_t* turns, int32_t batch,
                         int32_t mode, float* out);

/* Environment::place_stone (environment/src/lib.rs:104-166) on `batch` caller-held environments: boards [B][N*N] Stone bytes,
 * turns [B], legal [B] (legal_move_count) are updated in place; status_out[b] = Option<GameStatus> (-1 = None: the cell is
 * occupied or out of range and environment b is left unchanged).  batch = 1 is the scalar call of the Rust API. */
int omok_env_place_stone(omok_engine* e, uint8_t* boards, uint8_t* turns, uint16_t* legal, const int32_t* actions,
                         int32_t batch, int32_t* status_out);

/* ---- self-play: G games x two agents (src/trainer.rs:81-205) ---------------------------- */
/* Agent::new for both agents of every game (alpha-zero/src/agent.rs:16-35): root policy = raw
 * evaluate_p of the empty board.  Also clears the replay buffer.  Every reset is one trainer iteration
 * (src/trainer.rs:74-93, fresh thread_rng draws): it takes RNG stream `episode` and advances the counter; the first reset
 * after omok_create is episode 0. */
int omok_selfplay_reset(omok_engine* e);
/* index of the RNG stream the NEXT omok_selfplay_reset uses (resuming a training run at iteration i: omok_set_episode(e, i)) */
int omok_set_episode(omok_engine* e, uint64_t episode);
/* ParallelMCTSExecutor::execute (alpha-zero/src/parallel_mcts_executor.rs:26-35) on the
 * side-to-move agents of all live games: rounds of `batch_size` simulations per tree, one net
 * forward per round, ordered scatter; simulations round up to a multiple of batch_size. */
int omok_execute(omok_engine* e, int32_t count, int32_t batch_size, float epsilon, float alpha);
/* MCTSExecutor::run (alpha-zero/src/mcts_executor.rs:29-255; the executor of gui/src/agent.rs and benchmark/src/agent.rs) on an
 * engine with games = 1: ONE tree searched by `waves` wavefronts.  The reference runs its ceil(count / batch_size) rounds as
 * rayon tasks on one shared tree (relaxed atomics on n / w, the children lock in expand(), a duplicate expansion returns None
 * and drops the simulation, :171-178); here `waves` rounds run concurrently as the wavefronts of one workgroup, their requests
 * are evaluated as one batch and scattered by the same wavefronts.  waves = 1 is the sequential schedule: identical, bit for
 * bit, to omok_execute.  With waves > 1 the result depends on the interleaving, as it does in the reference. */
int omok_execute_shared(omok_engine* e, int32_t count, int32_t batch_size, float epsilon, float alpha, int32_t waves);
/* The same search under a RECORDED interleaving, for exact parity tests of waves > 1 (the reference's schedule is whatever its thread pool
 * does): every whole simulation and every backup of a scatter phase runs under the tree lock, i.e. the run is a sequential interleaving of
 * the waves' simulations -- one of the schedules the reference can produce -- and the lock order is reported so that a CPU restatement of
 * MCTSExecutor::run (mcts_executor.rs:76-255) can replay it.  G = ceil(ceil(count / batch_size) / waves) groups of `waves` rounds;
 * sim_order, backup_order [G][waves * batch_size]: wave index of the i-th simulation / backup of the group (0xFF beyond the group's
 * count); group_counts [G][3] = simulations, backups, requests; p [cap_requests][N*N], v [cap_requests]: the net outputs of all requests
 * in evaluation order (group by group, inside a group by wave then simulation); *n_groups, *n_requests. */
int omok_execute_shared_recorded(omok_engine* e, int32_t count, int32_t batch_size, float epsilon, float alpha, int32_t waves,
                                 uint8_t* sim_order, uint8_t* backup_order, int32_t* group_counts, float* p, float* v,
                                 int32_t cap_requests, int32_t* n_groups, int32_t* n_requests);
/* Agent::sample_action for every live game (agent.rs:83-137) with the trainer's mode rule
 * (trainer.rs:138-146): Boltzmann(temperature) while the game's ply < threshold, else Best.
 * Records the transition (env before the move, pi) like trainer.rs:150-173.
 * actions [G]: chosen cell, -1 for finished games.  May be NULL. */
int omok_sample_actions(omok_engine* e, float temperature, int32_t threshold, int32_t* actions);
/* Agent::play_action on the mover's tree, then ensure_action_exists + play_action on the
 * opponent's tree (agent.rs:144-232, trainer.rs:156-167), finished games retire
 * (trainer.rs:175-201).  Uses the actions chosen by the last omok_sample_actions. */
int omok_advance(omok_engine* e);
/* Agent::compute_policy (agent.rs:43-77) of the side-to-move agent of every game: pi [G][N*N] = child visit counts / their
 * sum; has_policy[g] = 0 where the reference returns None (finished game, no children, or no visits; the row is then 0).
 * has_policy may be NULL. */
int omok_compute_policy(omok_engine* e, float* pi, uint8_t* has_policy);
/* Externally chosen moves, actions [G] (ignored for finished games; every live game must move: all games share the side to
 * move, trainer.rs:96-97): Agent::ensure_action_exists(action) + Agent::play_action(action) on BOTH agents of each game
 * (agent.rs:144-232) -- what gui/src/agent.rs:49-66 and benchmark/src/agent.rs:34-50 do with a move their own search did not
 * pick.  One batched evaluate_p serves both agents of a game (same position).  No Transition is recorded (the trainer
 * records only moves it sampled, trainer.rs:138-173).  An occupied / out-of-range cell returns OMOK_ERR_ILLEGAL and leaves
 * every game unchanged (Option::None of play_action). */
int omok_play_actions(omok_engine* e, const int32_t* actions);
/* step-wise form for parity tests: stages the moves like omok_sample_actions does; omok_mirror_* / omok_advance follow */
int omok_set_actions(omok_engine* e, const int32_t* actions);
/* whole self-play phase of one trainer iteration (trainer.rs:95-205): repeats
 * execute/sample/advance until every game is finished or max_plies (>0) plies were played.
 * stats (may be NULL, OMOK_STAT_COUNT doubles): see OMOK_STAT_* */
int omok_selfplay_run(omok_engine* e, int32_t count, int32_t batch_size, float epsilon, float alpha,
                      float temperature, int32_t threshold, int32_t max_plies, double* stats);

/* Slots mode ("continuous refill"): plays `total_games` >= games games on the engine's `games` slots; a slot whose game is over takes the
   next game index instead of idling until the episode's longest game ends.  Per-game results are those of an episode of total_games
   games (omok_selfplay_run on an engine with games = total_games): a game's RNG streams are keyed by game_offset + index and its own
   ply, trees are independent (bit for bit with board_size 9 or OMOK_NET_F16X3_ROWS / OMOK_NET_F32; in the default net mode at
   board_size 15 a row's p / v carry ~5e-5 of rounding that depends on the path a round takes, see OMOK_NET_F16X3_ROWS).  Call after
   omok_selfplay_reset.  Finished games' transitions are appended to records_dev (device memory, cap_records records of
   omok_replay_record_bytes, the omok_replay_pack_dev format) in completion order; per game index: game_offsets[i] = first record,
   game_lengths[i] = records, game_status[i] = OMOK_STATUS_* (arrays of total_games, may be NULL); *n_records = records written.
   Extends src/trainer.rs:95-205 (the reference removes finished games from its agent list and lets the batch shrink). */
int omok_selfplay_run_slots(omok_engine* e, int32_t total_games, int32_t count, int32_t batch_size, float epsilon, float alpha,
                            float temperature, int32_t threshold, void* records_dev, int64_t cap_records, int64_t* game_offsets,
                            int32_t* game_lengths, int32_t* game_status, int64_t* n_records, double* stats);

/* step-wise form of execute() for parity tests: generate -> (eval | inject) -> scatter */
int omok_round_generate(omok_engine* e, int32_t round, int32_t batch_size, float epsilon, float alpha,
                        int32_t* n_requests);
int omok_round_inputs(omok_engine* e, float* inputs /* [n_requests][N][N][3] */);
int omok_round_eval(omok_engine* e);
int omok_round_outputs(omok_engine* e, float* p, float* v);
/* Precision evidence (tests, bench): the policy logits in front of the softmax [n_requests][HW] and the value in front of tanh [n_requests] (may be NULL)
 * of the round evaluated by omok_round_eval -- the quantities north_star's tolerance names (alpha-zero/src/network.rs:227-247), on the path the search
 * rounds take (sibling base + window differences).  Call between omok_round_eval and omok_round_scatter. */
int omok_round_logits(omok_engine* e, float* logits, float* vpre);
int omok_round_inject(omok_engine* e, const float* p, const float* v);
int omok_round_scatter(omok_engine* e);
/* step-wise form of the opponent-tree mirror eval inside omok_advance */
int omok_mirror_generate(omok_engine* e, int32_t* n_requests);
int omok_mirror_inputs(omok_engine* e, float* inputs);
int omok_mirror_eval(omok_engine* e);
int omok_mirror_outputs(omok_engine* e, float* p);
int omok_mirror_inject(omok_engine* e, const float* p);
int omok_mirror_apply(omok_engine* e);

/* ---- inspection ------------------------------------------------------------------------ */
int omok_alive_count(omok_engine* e);                  /* >= 0, or error */
int omok_current_ply(omok_engine* e);
int omok_game_info(omok_engine* e, uint8_t* alive, uint8_t* status, int32_t* plies); /* each [G], may be NULL */
/* canonical dump of one tree (MCTS::root / Node fields, mcts/src/node.rs:10-21):
 * ints [n][8] = parent, action, status, turn, legal_move_count, children, n, order|has_policy<<16
 * floats [n][1+N*N] = w, policy row.  returns the node count (or -count if cap is too small). */
int omok_tree_dump(omok_engine* e, int32_t game, int32_t side, int32_t* ints, float* floats, int32_t cap_nodes);
int omok_tree_root(omok_engine* e, int32_t game, int32_t side, uint32_t* root_n, float* root_w,
                   int32_t* n_nodes, int32_t* n_tables);
/* Node::children of a root in insertion order (mcts/src/node.rs:10-21; MCTS::root, mcts/src/lib.rs:34-36): action, n, w and
 * p (= root.policy[action], which the reference keeps equal to child.p) of the first min(children, cap) children.  Returns
 * the number of children.  Any output may be NULL. */
int omok_root_children(omok_engine* e, int32_t game, int32_t side, int32_t* actions, uint32_t* n, float* w, float* p, int32_t cap);
/* Transition{env, policy, z} records of one game (trainer.rs:20-24,169-173; z as recorded at play
 * time, before the back-fill of trainer.rs:207-214).  returns the ply count. */
int omok_replay_game(omok_engine* e, int32_t game, uint8_t* boards, uint8_t* turns, float* pi, float* z,
                     int32_t cap_plies);
/* replay tuples of all games packed on the device for an RCCL gather: record = board u8[N*N],
 * turn u8, zero pad to 4, pi f32[N*N], z f32; games in id order, transitions in play order (the same bytes on every run).
 * Writes at most cap_records to dst_dev (a device pointer the caller owns, e.g. a torch tensor) and returns the record
 * count. */
int64_t omok_replay_pack_dev(omok_engine* e, void* dst_dev, int64_t cap_records);
int32_t omok_replay_record_bytes(const omok_engine* e);
/* Replay post-processing of Trainer::train (src/trainer.rs:207-324) on the device.  Per game, in game-id order (the
 * reference walks `transitions` by game index too, trainer.rs:208): the game's L transitions with z back-filled (walking backwards from the
 * last transition z alternates sign, :209-214), then 5L augmented copies, transition-major, in the reference's order
 * rotate_90, rotate_180, rotate_270, flip_horizontal, flip_vertical of board and policy (src/utils.rs:1-64), turn and z
 * unchanged.  Records as in omok_replay_pack_dev.  Returns the record count 6 * sum(L) (records beyond cap are dropped). */
int64_t omok_replay_augment_dev(omok_engine* e, void* dst_dev, int64_t cap_records);
/* the same for one game into host arrays ([6L][N*N] boards / pi, [6L] turns / z); returns 6L */
int omok_replay_augmented_game(omok_engine* e, int32_t game, uint8_t* boards, uint8_t* turns, float* pi, float* z,
                               int32_t cap_records);

/* Debugging aid of the parity tests (no reference counterpart): the fc0 operand rows -- the trunk's output in the layout fc0 reads,
 * DESIGN 3.1 / 3.4 -- that the LAST forward left for request rows [first_row, first_row + rows), omok_operand_row_bytes each
 * (split-precision modes; on the copy path of sibling rounds they must equal the rows of a row-by-row evaluation bit for bit). */
int64_t omok_operand_row_bytes(const omok_engine* e);
int omok_debug_operand_rows(omok_engine* e, int32_t first_row, int32_t rows, void* out);
/* Debugging aid: enabled = 0 switches the base cache of the sibling rounds off (board_size 15, DESIGN 3.3: every run's base position is
 * then evaluated in full in every round instead of being kept while its leaf stays the tree's expansion target).  Results must not
 * change by a bit (tests). */
int omok_debug_set_base_cache(omok_engine* e, int32_t enabled);
/* Debugging aid / A-B switch: which kernel evaluates the children of a sibling run on the difference path: 2 = k_sib_children2 (default: one wave
 * per child, windows that grow with the blocks), 1 = k_sib_children (a wave pair per child, the 7x7 window through every block; always used on the copy path).
 * Outputs agree within 2e-4 (tests); cached base positions are dropped (the kernels read different base-slot layouts).  The environment variable
 * OMOK_SIB_V2=0 at omok_create selects 1 as the engine's default.  In the MIXED operand format (the usual outcome of omok_net_commit's probe) only k_sib_children2 writes
 * the fp6 difference rows: which = 1 then returns OMOK_ERR_STATE instead of silently changing nothing (use OMOK_NET_F16X3_FP6 / _F16 engines for an A-B run). */
int omok_debug_set_children_kernel(omok_engine* e, int32_t which);
/* Debugging aid: enabled = 0 makes the fc0 window tiles of sibling rounds walk the whole 7x7 window of their bin instead of the rectangle of window pixels their rows can
 * differ in (a child differs from its base only within its stone's pixel +- 3 clipped to the board; outside that region its difference row holds exact zeros).
 * Results must not change by a bit (tests): skipped pixels contribute exact zeros. */
int omok_debug_set_window_rects(omok_engine* e, int32_t enabled);

#define OMOK_STAT_SIMS 0        /* simulations run (incl. terminal hits / no-action sims) */
#define OMOK_STAT_EVALS 1       /* net evaluations (search requests + mirror evals + root) */
#define OMOK_STAT_PLY_GAMES 2   /* sum over plies of live games */
#define OMOK_STAT_FINISHED 3    /* games finished */
#define OMOK_STAT_MS_TREE 4     /* HIP-event ms in tree kernels (round+scan+scatter) */
#define OMOK_STAT_MS_TRUNK 5    /* ... net trunk kernel */
#define OMOK_STAT_MS_FC0 6      /* ... fc0 GEMM kernel */
#define OMOK_STAT_MS_TAIL 7     /* ... fc1/heads kernel */
#define OMOK_STAT_MS_PLY 8      /* ... sample/mirror/advance kernels */
#define OMOK_STAT_FC0_LAUNCHES 9
#define OMOK_STAT_FC0_ROWS 10   /* sum of batch rows over fc0 launches */
#define OMOK_STAT_TREE_BYTES 11 /* kernel-counted algorithmic bytes of the round kernels */
#define OMOK_STAT_ROUND_LAUNCHES 12
#define OMOK_STAT_MS_ROUND 13   /* HIP-event ms in the round (select/expand/backup) kernel only */
#define OMOK_STAT_PEAK_NODES 14  /* largest node / table arena use seen so far */
#define OMOK_STAT_PEAK_TABLES 15
#define OMOK_STAT_FC0_FORMAT 16  /* operand format of fc0's correction terms in use: 0 = block-scaled fp6, 1 = f16, 2 = mixed (f16 full rows, fp6 difference rows) (-1: OMOK_NET_F32) */
#define OMOK_STAT_PROBE_ROWS 17  /* rows of the last omok_net_commit's probe (0: no probe: forced format / OMOK_NET_F32) */
#define OMOK_STAT_PROBE_DP_FP6 18 /* the probe's max |dp|, |dv| against the fp32 kernels: fp6 correction terms ... */
#define OMOK_STAT_PROBE_DV_FP6 19
#define OMOK_STAT_PROBE_DP_F16 20 /* ... f16 correction terms */
#define OMOK_STAT_PROBE_DV_F16 21
#define OMOK_STAT_PROBE_LIMIT 22  /* fp6 is kept while both of its figures are <= this (3e-4) */
#define OMOK_STAT_PROBE_LOGIT_MAX 23 /* largest |policy logit| of the probe rows (fp32 kernels) */
#define OMOK_STAT_CHILDREN2_LAUNCHES 24 /* sibling rounds whose children ran on k_sib_children2 (difference path, default) ... */
#define OMOK_STAT_CHILDREN1_LAUNCHES 25 /* ... on k_sib_children (copy path; difference path after omok_debug_set_children_kernel(1)) */
#define OMOK_STAT_PROBE_DLOGIT_FP6 26 /* the probe's plain rows: max(|dlogit|, |dv before tanh|) against the fp32 kernels, fp6 / f16 correction terms */
#define OMOK_STAT_PROBE_DLOGIT_F16 27
#define OMOK_STAT_PROBE_ROUND_ROWS 28 /* rows of the probe's synthetic sibling round (difference path) that were checked against the fp32 kernels (0: this engine's
                                         rounds never take that path: games x max_batch_k below 3072 (board 15) / 1024 (board 9), or OMOK_NET_F16X3_ROWS) */
#define OMOK_STAT_PROBE_ROUND_FP6 29   /* [29..31] |dp|, |dv|, |dlogit| of that round with fp6 rows and fp6 difference rows */
#define OMOK_STAT_PROBE_ROUND_MIXED 32 /* [32..34] ... f16 rows, fp6 difference rows */
#define OMOK_STAT_PROBE_ROUND_F16 35   /* [35..37] ... f16 rows, f16 difference rows */
#define OMOK_STAT_PROBE_LOGIT_LIMIT 38 /* limit on the |dlogit| figures (5e-4); OMOK_STAT_PROBE_LIMIT (3e-4) is the one on |dp|, |dv| */
#define OMOK_STAT_PROBE_OUTSIDE 39 /* the probe's verdict on the format it committed: 0 = every figure inside the margin limits (OMOK_STAT_PROBE_LIMIT on |dp|, |dv|,
                                      OMOK_STAT_PROBE_LOGIT_LIMIT on the logits); 1 = the most precise split-operand format (f16 correction terms) is outside the margin but
                                      inside north_star's 1e-3 -- committed, one line on stderr; 2 = it is outside 1e-3: the engine evaluates this net with the plain fp32
                                      kernels (OMOK_STAT_FC0_FORMAT = -1, the arithmetic of agent_model.rs:116-134; slow) until the next omok_net_commit.  A forced format
                                      (OMOK_NET_F16X3_FP6 / _F16 / _MIXED) is never probed: 0 */
/* [40..49] executed work of the search rounds on the sibling paths since omok_reset_stats (bench.py's executed_flops): */
#define OMOK_STAT_WORK_DIFF_RUNS 40    /* difference path: runs of sibling requests ... */
#define OMOK_STAT_WORK_DIFF_SINGLES 41 /* ... request rows outside runs (evaluated in full) ... */
#define OMOK_STAT_WORK_DIFF_CHILDREN 42 /* ... request rows inside runs (k_sib_children2: a 5x5 / 7x7 window each) */
#define OMOK_STAT_WORK_COPY_RUNS 43    /* the same three on the copy path (k_trunk<BASE> per run, k_sib_children per row) */
#define OMOK_STAT_WORK_COPY_SINGLES 44
#define OMOK_STAT_WORK_COPY_CHILDREN 45
#define OMOK_STAT_WORK_DIFF_FULL_RUNS 46 /* runs of the difference path whose base was evaluated in full (base-cache misses + uncacheable runs) */
#define OMOK_STAT_WORK_WIN_PIXELS 47   /* window pixels walked by the fc0 window tiles (sum over tiles of their rectangles; K-split tiles: 49) */
#define OMOK_STAT_WORK_WIN_TILES 48    /* fc0 window tiles (128 slots each) */
#define OMOK_STAT_WORK_FULL_TILES 49   /* 128-row tiles of the difference path's full-row fc0 */
#define OMOK_STAT_COUNT 50
int omok_get_stats(omok_engine* e, double* stats /* [OMOK_STAT_COUNT] */);
int omok_reset_stats(omok_engine* e);
/* Per-category HIP-event timing of the kernels on the engine's stream (off by default).  enabled = 1: every launch; enabled = N > 1:
   search rounds are timed 1 in N and omok_get_stats scales the sampled sums by rounds seen / rounds timed (an event record costs the
   queue ~5 us, six category boundaries per round: 2 % of a full round, 15 % of a thin one); ply-level work is always timed. */
int omok_set_profiling(omok_engine* e, int32_t enabled);

#ifdef __cplusplus
}
#endif
#endif
