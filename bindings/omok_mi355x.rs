// omok_mi355x.rs -- Rust side of the C ABI of libomok_mi355x.so (include/omok_mi355x.h).
//
// What a maintainer of AcrylicShrimp/omok-ai adds as `alpha-zero/src/mi355x.rs` (`mod mi355x; pub use mi355x::*;` next to the
// modules of alpha-zero/src/lib.rs:1-17) to run the self-play phase of `Trainer::train` (src/trainer.rs:95-205) on an MI355X.
// The raw declarations (`ffi`) are GENERATED from the header by tools/gen_rust_binding.py; tests/test_abi.py checks every
// declaration against the header (names, argument count, integer widths, pointer constness, the layout of OmokConfig, constants).
// rustc / cargo are absent from the build image: this file is UNVERIFIED TEXT as far as the Rust compiler is concerned; the
// verified callers of the same ABI are tests/c/harness.c (C) and omok-ai_amd/binding.py (Python / ctypes).
//
// build.rs of the crate:   println!("cargo:rustc-link-search=native={}", "<repo>/omok-ai_amd");
//                          println!("cargo:rustc-link-lib=dylib=omok_mi355x");
#![allow(non_camel_case_types, dead_code)]

use std::ffi::{CStr, CString};
use std::os::raw::{c_char, c_int, c_void};
use std::ptr;

/// `omok_config` (include/omok_mi355x.h): field for field, `#[repr(C)]`.
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct OmokConfig {
    pub board_size: i32,     // Environment::BOARD_SIZE (environment/src/lib.rs:70): 9 or 15
    pub games: i32,          // episode_count (src/config.rs:90); two trees (agents) per game
    pub max_nodes: i32,      // per-tree node arena (replaces BumpAllocator's pages, mcts/src/bump_allocator.rs:7-60)
    pub max_tables: i32,     // per-tree child-table arena
    pub max_batch_k: i32,    // largest evaluate_batch_size (src/config.rs:92)
    pub device: i32,         // HIP device ordinal
    pub net_mode: i32,       // OMOK_NET_*: 0 split-operand MFMA (fc0 format chosen by omok_net_commit's probe), 1 fp32 kernels, 2 as 0 with every
                             // request row on its own, 3 / 4 / 5 as 0 with fc0's correction terms forced to fp6 / f16 / mixed
    pub max_tree_waves: i32, // 0, or the largest `waves` of omok_execute_shared (<= 16)
    pub seed: u64,           // the reference draws from thread_rng(); here the stream is (seed, episode, game_offset)
    pub game_offset: i64,    // global id of game 0: rank * games when the games are sharded over GPUs
}

/// Opaque `omok_engine`.
#[repr(C)]
pub struct OmokEngine {
    _private: [u8; 0],
}

// ---- constants of the header (generated) ----
pub const OMOK_OK: i32 = 0;
pub const OMOK_ERR_INVALID: i32 = -1;
pub const OMOK_ERR_HIP: i32 = -2;
pub const OMOK_ERR_STATE: i32 = -3;
pub const OMOK_ERR_OVERFLOW: i32 = -4;
pub const OMOK_ERR_ILLEGAL: i32 = -5;
pub const OMOK_MODE_PLAYER: i32 = 0;
pub const OMOK_MODE_OPPONENT: i32 = 1;
pub const OMOK_NET_F16X3: i32 = 0;
pub const OMOK_NET_F32: i32 = 1;
pub const OMOK_NET_F16X3_ROWS: i32 = 2;
pub const OMOK_NET_F16X3_FP6: i32 = 3;
pub const OMOK_NET_F16X3_F16: i32 = 4;
pub const OMOK_NET_F16X3_MIXED: i32 = 5;
pub const OMOK_MAX_ARENA: i32 = 16384;
pub const OMOK_STAT_SIMS: i32 = 0;
pub const OMOK_STAT_EVALS: i32 = 1;
pub const OMOK_STAT_PLY_GAMES: i32 = 2;
pub const OMOK_STAT_FINISHED: i32 = 3;
pub const OMOK_STAT_MS_TREE: i32 = 4;
pub const OMOK_STAT_MS_TRUNK: i32 = 5;
pub const OMOK_STAT_MS_FC0: i32 = 6;
pub const OMOK_STAT_MS_TAIL: i32 = 7;
pub const OMOK_STAT_MS_PLY: i32 = 8;
pub const OMOK_STAT_FC0_LAUNCHES: i32 = 9;
pub const OMOK_STAT_FC0_ROWS: i32 = 10;
pub const OMOK_STAT_TREE_BYTES: i32 = 11;
pub const OMOK_STAT_ROUND_LAUNCHES: i32 = 12;
pub const OMOK_STAT_MS_ROUND: i32 = 13;
pub const OMOK_STAT_PEAK_NODES: i32 = 14;
pub const OMOK_STAT_PEAK_TABLES: i32 = 15;
pub const OMOK_STAT_FC0_FORMAT: i32 = 16;
pub const OMOK_STAT_PROBE_ROWS: i32 = 17;
pub const OMOK_STAT_PROBE_DP_FP6: i32 = 18;
pub const OMOK_STAT_PROBE_DV_FP6: i32 = 19;
pub const OMOK_STAT_PROBE_DP_F16: i32 = 20;
pub const OMOK_STAT_PROBE_DV_F16: i32 = 21;
pub const OMOK_STAT_PROBE_LIMIT: i32 = 22;
pub const OMOK_STAT_PROBE_LOGIT_MAX: i32 = 23;
pub const OMOK_STAT_CHILDREN2_LAUNCHES: i32 = 24;
pub const OMOK_STAT_CHILDREN1_LAUNCHES: i32 = 25;
pub const OMOK_STAT_PROBE_DLOGIT_FP6: i32 = 26;
pub const OMOK_STAT_PROBE_DLOGIT_F16: i32 = 27;
pub const OMOK_STAT_PROBE_ROUND_ROWS: i32 = 28;
pub const OMOK_STAT_PROBE_ROUND_FP6: i32 = 29;
pub const OMOK_STAT_PROBE_ROUND_MIXED: i32 = 32;
pub const OMOK_STAT_PROBE_ROUND_F16: i32 = 35;
pub const OMOK_STAT_PROBE_LOGIT_LIMIT: i32 = 38;
pub const OMOK_STAT_PROBE_OUTSIDE: i32 = 39;
pub const OMOK_STAT_WORK_DIFF_RUNS: i32 = 40;
pub const OMOK_STAT_WORK_DIFF_SINGLES: i32 = 41;
pub const OMOK_STAT_WORK_DIFF_CHILDREN: i32 = 42;
pub const OMOK_STAT_WORK_COPY_RUNS: i32 = 43;
pub const OMOK_STAT_WORK_COPY_SINGLES: i32 = 44;
pub const OMOK_STAT_WORK_COPY_CHILDREN: i32 = 45;
pub const OMOK_STAT_WORK_DIFF_FULL_RUNS: i32 = 46;
pub const OMOK_STAT_WORK_WIN_PIXELS: i32 = 47;
pub const OMOK_STAT_WORK_WIN_TILES: i32 = 48;
pub const OMOK_STAT_WORK_FULL_TILES: i32 = 49;
pub const OMOK_STAT_COUNT: i32 = 50;

/// The raw C ABI: every entry point of include/omok_mi355x.h (generated; do not edit by hand).
pub mod ffi {
    use super::{OmokConfig, OmokEngine};
    use std::os::raw::{c_char, c_int, c_void};

    #[link(name = "omok_mi355x")]
    extern "C" {
        pub fn omok_create(cfg: *const OmokConfig, out: *mut *mut OmokEngine) -> c_int;
        pub fn omok_destroy(e: *mut OmokEngine);
        pub fn omok_last_error(e: *const OmokEngine) -> *const c_char;
        pub fn omok_net_num_tensors() -> c_int;
        pub fn omok_net_tensor_size(e: *const OmokEngine, index: c_int) -> i64;
        pub fn omok_net_load(e: *mut OmokEngine, index: c_int, data: *const f32, count: i64) -> c_int;
        pub fn omok_net_commit(e: *mut OmokEngine) -> c_int;
        pub fn omok_net_load_file(e: *mut OmokEngine, path: *const c_char) -> c_int;
        pub fn omok_net_save_file(e: *mut OmokEngine, path: *const c_char) -> c_int;
        pub fn omok_evaluate_pv(e: *mut OmokEngine, input: *const f32, batch: i32, p: *mut f32, v: *mut f32) -> c_int;
        pub fn omok_evaluate_logits(e: *mut OmokEngine, input: *const f32, batch: i32, logits: *mut f32, vpre: *mut f32) -> c_int;
        pub fn omok_env_play(e: *mut OmokEngine, moves: *const i32, batch: i32, len: i32, status_out: *mut i32, boards_out: *mut u8, turns_out: *mut u8, legal_out: *mut u16) -> c_int;
        pub fn omok_encode_nn_input(e: *mut OmokEngine, boards: *const u8, turns: *const u8, batch: i32, mode: i32, out: *mut f32) -> c_int;
        pub fn omok_env_place_stone(e: *mut OmokEngine, boards: *mut u8, turns: *mut u8, legal: *mut u16, actions: *const i32, batch: i32, status_out: *mut i32) -> c_int;
        pub fn omok_selfplay_reset(e: *mut OmokEngine) -> c_int;
        pub fn omok_set_episode(e: *mut OmokEngine, episode: u64) -> c_int;
        pub fn omok_execute(e: *mut OmokEngine, count: i32, batch_size: i32, epsilon: f32, alpha: f32) -> c_int;
        pub fn omok_execute_shared(e: *mut OmokEngine, count: i32, batch_size: i32, epsilon: f32, alpha: f32, waves: i32) -> c_int;
        pub fn omok_execute_shared_recorded(e: *mut OmokEngine, count: i32, batch_size: i32, epsilon: f32, alpha: f32, waves: i32, sim_order: *mut u8, backup_order: *mut u8, group_counts: *mut i32, p: *mut f32, v: *mut f32, cap_requests: i32, n_groups: *mut i32, n_requests: *mut i32) -> c_int;
        pub fn omok_sample_actions(e: *mut OmokEngine, temperature: f32, threshold: i32, actions: *mut i32) -> c_int;
        pub fn omok_advance(e: *mut OmokEngine) -> c_int;
        pub fn omok_compute_policy(e: *mut OmokEngine, pi: *mut f32, has_policy: *mut u8) -> c_int;
        pub fn omok_play_actions(e: *mut OmokEngine, actions: *const i32) -> c_int;
        pub fn omok_set_actions(e: *mut OmokEngine, actions: *const i32) -> c_int;
        pub fn omok_selfplay_run(e: *mut OmokEngine, count: i32, batch_size: i32, epsilon: f32, alpha: f32, temperature: f32, threshold: i32, max_plies: i32, stats: *mut f64) -> c_int;
        pub fn omok_selfplay_run_slots(e: *mut OmokEngine, total_games: i32, count: i32, batch_size: i32, epsilon: f32, alpha: f32, temperature: f32, threshold: i32, records_dev: *mut c_void, cap_records: i64, game_offsets: *mut i64, game_lengths: *mut i32, game_status: *mut i32, n_records: *mut i64, stats: *mut f64) -> c_int;
        pub fn omok_round_generate(e: *mut OmokEngine, round: i32, batch_size: i32, epsilon: f32, alpha: f32, n_requests: *mut i32) -> c_int;
        pub fn omok_round_inputs(e: *mut OmokEngine, inputs: *mut f32) -> c_int;
        pub fn omok_round_eval(e: *mut OmokEngine) -> c_int;
        pub fn omok_round_outputs(e: *mut OmokEngine, p: *mut f32, v: *mut f32) -> c_int;
        pub fn omok_round_logits(e: *mut OmokEngine, logits: *mut f32, vpre: *mut f32) -> c_int;
        pub fn omok_round_inject(e: *mut OmokEngine, p: *const f32, v: *const f32) -> c_int;
        pub fn omok_round_scatter(e: *mut OmokEngine) -> c_int;
        pub fn omok_mirror_generate(e: *mut OmokEngine, n_requests: *mut i32) -> c_int;
        pub fn omok_mirror_inputs(e: *mut OmokEngine, inputs: *mut f32) -> c_int;
        pub fn omok_mirror_eval(e: *mut OmokEngine) -> c_int;
        pub fn omok_mirror_outputs(e: *mut OmokEngine, p: *mut f32) -> c_int;
        pub fn omok_mirror_inject(e: *mut OmokEngine, p: *const f32) -> c_int;
        pub fn omok_mirror_apply(e: *mut OmokEngine) -> c_int;
        pub fn omok_alive_count(e: *mut OmokEngine) -> c_int;
        pub fn omok_current_ply(e: *mut OmokEngine) -> c_int;
        pub fn omok_game_info(e: *mut OmokEngine, alive: *mut u8, status: *mut u8, plies: *mut i32) -> c_int;
        pub fn omok_tree_dump(e: *mut OmokEngine, game: i32, side: i32, ints: *mut i32, floats: *mut f32, cap_nodes: i32) -> c_int;
        pub fn omok_tree_root(e: *mut OmokEngine, game: i32, side: i32, root_n: *mut u32, root_w: *mut f32, n_nodes: *mut i32, n_tables: *mut i32) -> c_int;
        pub fn omok_root_children(e: *mut OmokEngine, game: i32, side: i32, actions: *mut i32, n: *mut u32, w: *mut f32, p: *mut f32, cap: i32) -> c_int;
        pub fn omok_replay_game(e: *mut OmokEngine, game: i32, boards: *mut u8, turns: *mut u8, pi: *mut f32, z: *mut f32, cap_plies: i32) -> c_int;
        pub fn omok_replay_pack_dev(e: *mut OmokEngine, dst_dev: *mut c_void, cap_records: i64) -> i64;
        pub fn omok_replay_record_bytes(e: *const OmokEngine) -> i32;
        pub fn omok_replay_augment_dev(e: *mut OmokEngine, dst_dev: *mut c_void, cap_records: i64) -> i64;
        pub fn omok_replay_augmented_game(e: *mut OmokEngine, game: i32, boards: *mut u8, turns: *mut u8, pi: *mut f32, z: *mut f32, cap_records: i32) -> c_int;
        pub fn omok_operand_row_bytes(e: *const OmokEngine) -> i64;
        pub fn omok_debug_operand_rows(e: *mut OmokEngine, first_row: i32, rows: i32, out: *mut c_void) -> c_int;
        pub fn omok_debug_set_base_cache(e: *mut OmokEngine, enabled: i32) -> c_int;
        pub fn omok_debug_set_children_kernel(e: *mut OmokEngine, which: i32) -> c_int;
        pub fn omok_debug_set_window_rects(e: *mut OmokEngine, enabled: i32) -> c_int;
        pub fn omok_get_stats(e: *mut OmokEngine, stats: *mut f64) -> c_int;
        pub fn omok_reset_stats(e: *mut OmokEngine) -> c_int;
        pub fn omok_set_profiling(e: *mut OmokEngine, enabled: i32) -> c_int;
    }
}

// ------------------------------------------------------------------------------------------------
// Safe wrappers with the names of the crate APIs they stand in for.
// ------------------------------------------------------------------------------------------------

/// `GameStatus` in the reference's declaration order (environment/src/lib.rs:46-51).
#[repr(u8)]
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub enum GameStatus {
    InProgress = 0,
    Draw = 1,
    BlackWin = 2,
    WhiteWin = 3,
}

impl GameStatus {
    pub fn from_raw(v: i32) -> Option<GameStatus> {
        match v {
            0 => Some(GameStatus::InProgress),
            1 => Some(GameStatus::Draw),
            2 => Some(GameStatus::BlackWin),
            3 => Some(GameStatus::WhiteWin),
            _ => None, // -1 = Option::None of the reference
        }
    }
}

/// What `Result<_, tensorflow::Status>` becomes: the negative return code + `omok_last_error`.
#[derive(Debug)]
pub struct OmokError {
    pub code: i32,
    pub message: String,
}

/// One engine = one GPU = the `Session` + `AgentModel` + `ParallelMCTSExecutor` + both `Vec<Agent>` of `Trainer::train`
/// (src/trainer.rs:35-48,81-93).  `Send` but not `Sync`: one host thread per GPU.
pub struct Engine {
    raw: *mut OmokEngine,
    hw: usize,
    games: usize,
}
unsafe impl Send for Engine {}

impl Engine {
    fn check(&self, rc: c_int) -> Result<c_int, OmokError> {
        if rc >= 0 {
            return Ok(rc);
        }
        let message = unsafe { CStr::from_ptr(ffi::omok_last_error(self.raw)) }.to_string_lossy().into_owned();
        Err(OmokError { code: rc, message })
    }

    /// `Trainer::new` (src/trainer.rs:35-48).
    pub fn new(cfg: &OmokConfig) -> Result<Engine, OmokError> {
        let mut raw: *mut OmokEngine = ptr::null_mut();
        let rc = unsafe { ffi::omok_create(cfg, &mut raw) };
        if rc < 0 {
            let message = unsafe { CStr::from_ptr(ffi::omok_last_error(ptr::null())) }.to_string_lossy().into_owned();
            return Err(OmokError { code: rc, message });
        }
        Ok(Engine { raw, hw: (cfg.board_size * cfg.board_size) as usize, games: cfg.games as usize })
    }

    /// `ModelIO::load` (alpha-zero/src/model_io.rs:92-120) on the reference's `saves/<model_name>` file.
    pub fn load(&mut self, path: &str) -> Result<(), OmokError> {
        let c = CString::new(path).unwrap();
        self.check(unsafe { ffi::omok_net_load_file(self.raw, c.as_ptr()) }).map(|_| ())
    }

    /// `ModelIO::save` (model_io.rs:59-90).
    pub fn save(&mut self, path: &str) -> Result<(), OmokError> {
        let c = CString::new(path).unwrap();
        self.check(unsafe { ffi::omok_net_save_file(self.raw, c.as_ptr()) }).map(|_| ())
    }

    /// The 31 variables of `Network::variables` in order (network.rs:78-79,113-122,149-150,162-163,201-202,240-241), then commit.
    pub fn load_tensors(&mut self, tensors: &[Vec<f32>]) -> Result<(), OmokError> {
        for (i, t) in tensors.iter().enumerate() {
            self.check(unsafe { ffi::omok_net_load(self.raw, i as c_int, t.as_ptr(), t.len() as i64) })?;
        }
        self.check(unsafe { ffi::omok_net_commit(self.raw) }).map(|_| ())
    }

    /// `AgentModel::evaluate_pv` (agent_model.rs:116-134): input `[B, N, N, 3]` (encoder.rs:10-46) -> (p `[B, N*N]`, v `[B]`).
    pub fn evaluate_pv(&mut self, input: &[f32]) -> Result<(Vec<f32>, Vec<f32>), OmokError> {
        let batch = input.len() / (3 * self.hw);
        let (mut p, mut v) = (vec![0f32; batch * self.hw], vec![0f32; batch]);
        self.check(unsafe { ffi::omok_evaluate_pv(self.raw, input.as_ptr(), batch as i32, p.as_mut_ptr(), v.as_mut_ptr()) })?;
        Ok((p, v))
    }

    /// `Environment::place_stone` (environment/src/lib.rs:104-166) on one caller-held environment; `None` = occupied cell.
    pub fn place_stone(&mut self, board: &mut [u8], turn: &mut u8, legal_move_count: &mut u16, index: usize) -> Result<Option<GameStatus>, OmokError> {
        let (action, mut status) = (index as i32, -1i32);
        self.check(unsafe { ffi::omok_env_place_stone(self.raw, board.as_mut_ptr(), turn, legal_move_count, &action, 1, &mut status) })?;
        Ok(GameStatus::from_raw(status))
    }

    /// `Agent::new` for both agents of every game (agent.rs:16-35, trainer.rs:89-93).
    pub fn selfplay_reset(&mut self) -> Result<(), OmokError> {
        self.check(unsafe { ffi::omok_selfplay_reset(self.raw) }).map(|_| ())
    }

    /// `ParallelMCTSExecutor::execute` (parallel_mcts_executor.rs:26-35) on the side-to-move agents (trainer.rs:99-122).
    pub fn execute(&mut self, count: usize, batch_size: usize, epsilon: f32, alpha: f32) -> Result<(), OmokError> {
        self.check(unsafe { ffi::omok_execute(self.raw, count as i32, batch_size as i32, epsilon, alpha) }).map(|_| ())
    }

    /// `MCTSExecutor::run` (mcts_executor.rs:29-255) on an engine with `games = 1`.
    pub fn execute_shared(&mut self, count: usize, batch_size: usize, epsilon: f32, alpha: f32, waves: usize) -> Result<(), OmokError> {
        self.check(unsafe { ffi::omok_execute_shared(self.raw, count as i32, batch_size as i32, epsilon, alpha, waves as i32) }).map(|_| ())
    }

    /// `Agent::sample_action` of every live game with the trainer's mode rule (agent.rs:83-137, trainer.rs:138-146); -1 = finished game.
    pub fn sample_actions(&mut self, temperature: f32, temperature_threshold: usize) -> Result<Vec<i32>, OmokError> {
        let mut actions = vec![-1i32; self.games];
        self.check(unsafe { ffi::omok_sample_actions(self.raw, temperature, temperature_threshold as i32, actions.as_mut_ptr()) })?;
        Ok(actions)
    }

    /// `play_action` on the mover, `ensure_action_exists` + `play_action` on the opponent, retire finished games (trainer.rs:156-201).
    pub fn advance(&mut self) -> Result<(), OmokError> {
        self.check(unsafe { ffi::omok_advance(self.raw) }).map(|_| ())
    }

    /// `Agent::compute_policy` (agent.rs:43-77) for every game; `None` where the reference returns `None`.
    pub fn compute_policy(&mut self) -> Result<Vec<Option<Vec<f32>>>, OmokError> {
        let (mut pi, mut has) = (vec![0f32; self.games * self.hw], vec![0u8; self.games]);
        self.check(unsafe { ffi::omok_compute_policy(self.raw, pi.as_mut_ptr(), has.as_mut_ptr()) })?;
        Ok((0..self.games).map(|g| if has[g] != 0 { Some(pi[g * self.hw..(g + 1) * self.hw].to_vec()) } else { None }).collect())
    }

    /// `Agent::ensure_action_exists` + `Agent::play_action` with moves chosen outside the engine (agent.rs:144-232; gui / benchmark).
    /// `Ok(false)` = `None` of `play_action` (occupied cell): nothing changed.
    pub fn play_actions(&mut self, actions: &[i32]) -> Result<bool, OmokError> {
        let rc = unsafe { ffi::omok_play_actions(self.raw, actions.as_ptr()) };
        if rc == OMOK_ERR_ILLEGAL {
            return Ok(false);
        }
        self.check(rc).map(|_| true)
    }

    /// The whole `while !agents_1.is_empty()` loop of `Trainer::train` (trainer.rs:95-205); returns the statistics (`OMOK_STAT_*`).
    pub fn selfplay_run(&mut self, count: usize, batch_size: usize, epsilon: f32, alpha: f32, temperature: f32, temperature_threshold: usize)
                        -> Result<Vec<f64>, OmokError> {
        let mut stats = vec![0f64; OMOK_STAT_COUNT as usize];
        self.check(unsafe {
            ffi::omok_selfplay_run(self.raw, count as i32, batch_size as i32, epsilon, alpha, temperature, temperature_threshold as i32, 0, stats.as_mut_ptr())
        })?;
        Ok(stats)
    }

    pub fn alive_count(&mut self) -> Result<usize, OmokError> {
        self.check(unsafe { ffi::omok_alive_count(self.raw) }).map(|n| n as usize)
    }

    /// `Transition { env, policy, z }` of one game (trainer.rs:20-24,169-173): (boards, turns, policies, z) per ply.
    pub fn replay_game(&mut self, game: usize) -> Result<(Vec<u8>, Vec<u8>, Vec<f32>, Vec<f32>), OmokError> {
        let cap = self.hw;
        let (mut b, mut t, mut pi, mut z) = (vec![0u8; cap * self.hw], vec![0u8; cap], vec![0f32; cap * self.hw], vec![0f32; cap]);
        let n = self.check(unsafe {
            ffi::omok_replay_game(self.raw, game as i32, b.as_mut_ptr(), t.as_mut_ptr(), pi.as_mut_ptr(), z.as_mut_ptr(), cap as i32)
        })? as usize;
        b.truncate(n * self.hw);
        t.truncate(n);
        pi.truncate(n * self.hw);
        z.truncate(n);
        Ok((b, t, pi, z))
    }

    /// z back-fill + the five augmentations of trainer.rs:207-324, packed on the device (records of `omok_replay_record_bytes`).
    pub fn replay_augment_dev(&mut self, dst_dev: *mut c_void, cap_records: i64) -> i64 {
        unsafe { ffi::omok_replay_augment_dev(self.raw, dst_dev, cap_records) }
    }

    pub fn stats(&mut self) -> Result<Vec<f64>, OmokError> {
        let mut stats = vec![0f64; OMOK_STAT_COUNT as usize];
        self.check(unsafe { ffi::omok_get_stats(self.raw, stats.as_mut_ptr()) })?;
        Ok(stats)
    }

    pub fn raw(&mut self) -> *mut OmokEngine {
        self.raw
    }
}

impl Drop for Engine {
    fn drop(&mut self) {
        unsafe { ffi::omok_destroy(self.raw) }
    }
}

// The self-play phase of one iteration, as src/trainer.rs:95-205 reads with the engine in place of its inner loops:
//
//     engine.selfplay_reset()?;                                        // Agent::new x 2 per game        (trainer.rs:89-93)
//     while engine.alive_count()? > 0 {
//         engine.execute(evaluate_count, evaluate_batch_size, epsilon, alpha)?;   // execute(...)         (trainer.rs:99-122)
//         engine.sample_actions(temperature, temperature_threshold)?;  // sample_action per game         (trainer.rs:138-146)
//         engine.advance()?;                                           // play_action / ensure_action_exists / swap_remove (:156-201)
//     }
//     for game in 0..episode_count { let (boards, turns, pi, z) = engine.replay_game(game)?; /* trainer.rs:207-324 as it is */ }
