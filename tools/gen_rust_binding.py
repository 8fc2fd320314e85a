"""Writes bindings/omok_mi355x.rs from include/omok_mi355x.h: constants and the complete `extern "C"` block are generated from the
header's prototypes (tools/abi_text.py), the safe wrappers around them are the fixed text below.  Run after every change of the header:
    python tools/gen_rust_binding.py
tests/test_abi.py compares the committed file with the header (names, arity, integer widths, constness, struct layout, constants)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import abi_text as A  # noqa: E402

HEAD = '''// omok_mi355x.rs -- Rust side of the C ABI of libomok_mi355x.so (include/omok_mi355x.h).
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

'''

WRAPPERS = '''
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
'''


def main():
    sigs = A.parse_header()
    out = [HEAD]
    out.append("// ---- constants of the header (generated) ----\n")
    for k, v in A.parse_defines().items():
        out.append(f"pub const {k}: i32 = {v};\n")
    out.append('\n/// The raw C ABI: every entry point of include/omok_mi355x.h (generated; do not edit by hand).\npub mod ffi {\n')
    out.append("    use super::{OmokConfig, OmokEngine};\n    use std::os::raw::{c_char, c_int, c_void};\n\n")
    out.append('    #[link(name = "omok_mi355x")]\n    extern "C" {\n')
    for name, sig in sigs.items():
        out.append("    " + A.rust_decl(name, sig) + "\n")
    out.append("    }\n}\n")
    out.append(WRAPPERS)
    os.makedirs(os.path.dirname(A.RUST), exist_ok=True)
    with open(A.RUST, "w") as f:
        f.write("".join(out))
    print(f"wrote {A.RUST}: {len(sigs)} entry points")


if __name__ == "__main__":
    main()
