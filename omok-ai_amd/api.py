"""Host-side mirror of the reference's crate API for the self-play path, over the C ABI.

Names follow the reference so parity tests read like its own code:
  Environment                      environment/src/lib.rs:62-166
  encode_nn_input                  alpha-zero/src/encoder.rs:10-46
  AgentModel.evaluate_p/_pv        alpha-zero/src/agent_model.rs:105-134
  SelfPlay.execute                 ParallelMCTSExecutor::execute, alpha-zero/src/parallel_mcts_executor.rs:26-35
  SelfPlay.sample_actions/advance  Agent::{sample_action, play_action, ensure_action_exists}, agent.rs:83-232
  SelfPlay.run                     Trainer::train self-play phase, src/trainer.rs:95-205
All compute happens in the HIP library; nothing here has a CPU path.
"""
import ctypes as C

import os

import numpy as np

from . import binding as B
from . import weights as W

EMPTY, BLACK, WHITE = 0, 1, 2
TURN_BLACK, TURN_WHITE = 0, 1
IN_PROGRESS, DRAW, BLACK_WIN, WHITE_WIN = 0, 1, 2, 3


class Engine:
    """Opaque engine handle (omok_create / omok_destroy)."""

    def __init__(self, board_size=15, games=1, max_nodes=2048, max_tables=1024, max_batch_k=16, device=0,
                 net_mode=B.NET_F16X3, seed=0, game_offset=0, max_tree_waves=0):
        self.n, self.hw, self.games = board_size, board_size * board_size, games
        self.max_nodes = max_nodes
        self.max_batch_k = max_batch_k
        cfg = B.Config(board_size, games, max_nodes, max_tables, max_batch_k, device, net_mode, max_tree_waves, seed, game_offset)
        h = C.c_void_p()
        rc = B.lib().omok_create(C.byref(cfg), C.byref(h))
        if rc != 0:
            raise B.OmokError(rc, B.lib().omok_last_error(None).decode())
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            B.lib().omok_destroy(self.h)
            self.h = None

    def __del__(self):
        try:  # (at interpreter shutdown the module globals may already be gone: the process is ending, the driver frees the device memory)
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc < 0:
            raise B.OmokError(rc, B.lib().omok_last_error(self.h).decode())
        return rc

    # ---- net --------------------------------------------------------------------------------
    def load_weights(self, tensors):
        """31 tensors in the reference's variable order (network.rs / model_io.rs positional order)."""
        assert len(tensors) == B.lib().omok_net_num_tensors()
        for i, t in enumerate(tensors):
            t = np.ascontiguousarray(t, dtype=np.float32).ravel()
            self._chk(B.lib().omok_net_load(self.h, i, B.fptr(t), t.size))
        self._chk(B.lib().omok_net_commit(self.h))

    def load(self, path):
        """ModelIO::load (alpha-zero/src/model_io.rs:92-120): the reference's bincode weights file, positional."""
        self._chk(B.lib().omok_net_load_file(self.h, os.fsencode(path)))

    def save(self, path):
        """ModelIO::save (alpha-zero/src/model_io.rs:59-90)."""
        self._chk(B.lib().omok_net_save_file(self.h, os.fsencode(path)))

    def load_random_weights(self, seed=0):
        self.load_weights(W.init_random(self.n, seed))

    def evaluate_pv(self, inputs):
        x = np.ascontiguousarray(inputs, dtype=np.float32).reshape(-1, 3 * self.hw)
        b = x.shape[0]
        p = np.zeros((b, self.hw), dtype=np.float32)
        v = np.zeros(b, dtype=np.float32)
        self._chk(B.lib().omok_evaluate_pv(self.h, B.fptr(x), b, B.fptr(p), B.fptr(v)))
        return p.reshape(b, self.n, self.n), v.reshape(b, 1)

    def evaluate_logits(self, inputs):
        """pre-softmax policy logits [B][HW] and pre-tanh value [B] of the same forward (precision evidence)"""
        x = np.ascontiguousarray(inputs, dtype=np.float32).reshape(-1, 3 * self.hw)
        b = x.shape[0]
        lg = np.zeros((b, self.hw), dtype=np.float32)
        vp = np.zeros(b, dtype=np.float32)
        self._chk(B.lib().omok_evaluate_logits(self.h, B.fptr(x), b, B.fptr(lg), B.fptr(vp)))
        return lg, vp

    def evaluate_p(self, inputs):
        return self.evaluate_pv(inputs)[0]

    # ---- environment ------------------------------------------------------------------------
    def env_play(self, moves):
        """moves [B][L] int32 -> (status [B][L], boards [B][HW], turns [B], legal [B])."""
        moves = np.ascontiguousarray(moves, dtype=np.int32)
        if moves.ndim == 1:
            moves = moves[None]
        b, l = moves.shape
        status = np.zeros((b, max(l, 1)), dtype=np.int32)
        boards = np.zeros((b, self.hw), dtype=np.uint8)
        turns = np.zeros(b, dtype=np.uint8)
        legal = np.zeros(b, dtype=np.uint16)
        self._chk(B.lib().omok_env_play(self.h, B.iptr(moves), b, l, B.iptr(status), B.u8ptr(boards), B.u8ptr(turns),
                                        legal.ctypes.data_as(C.POINTER(C.c_uint16))))
        return status[:, :l], boards, turns, legal

    def env_place_stone(self, boards, turns, legal, actions):
        """Environment::place_stone on caller-held environments (updated in place); returns status [B] (-1 = None)."""
        b = len(actions)
        assert boards.dtype == np.uint8 and boards.shape == (b, self.hw) and boards.flags.c_contiguous
        assert turns.dtype == np.uint8 and legal.dtype == np.uint16
        actions = np.ascontiguousarray(actions, dtype=np.int32)
        status = np.zeros(b, dtype=np.int32)
        self._chk(B.lib().omok_env_place_stone(self.h, B.u8ptr(boards), B.u8ptr(turns), legal.ctypes.data_as(C.POINTER(C.c_uint16)),
                                               B.iptr(actions), b, B.iptr(status)))
        return status

    def encode_nn_input(self, boards, turns, mode=B.MODE_PLAYER):
        boards = np.ascontiguousarray(boards, dtype=np.uint8).reshape(-1, self.hw)
        turns = np.ascontiguousarray(turns, dtype=np.uint8).reshape(-1)
        out = np.zeros((boards.shape[0], 3 * self.hw), dtype=np.float32)
        self._chk(B.lib().omok_encode_nn_input(self.h, B.u8ptr(boards), B.u8ptr(turns), boards.shape[0], mode, B.fptr(out)))
        return out.reshape(-1, self.n, self.n, 3)

    # ---- stats ------------------------------------------------------------------------------
    def operand_rows(self, first_row, rows):
        """uint8 [rows][omok_operand_row_bytes]: the fc0 operand rows the last forward left (omok_debug_operand_rows)."""
        nb = int(B.lib().omok_operand_row_bytes(self.h))
        if nb < 0:
            raise B.OmokError(nb, "no operand rows in this net mode")
        out = np.empty((rows, nb), dtype=np.uint8)
        self._chk(B.lib().omok_debug_operand_rows(self.h, first_row, rows, out.ctypes.data_as(C.c_void_p)))
        return out

    def set_base_cache(self, on=True):
        self._chk(B.lib().omok_debug_set_base_cache(self.h, int(bool(on))))

    def set_window_rects(self, on=True):
        """False: fc0 window tiles walk their bin's whole 7x7 window instead of the rectangle their rows can differ in (same bits: tests)."""
        self._chk(B.lib().omok_debug_set_window_rects(self.h, int(bool(on))))

    def set_children_kernel(self, which=2):
        """2: k_sib_children2 (default), 1: k_sib_children on the difference path of sibling rounds (A-B / tests)."""
        self._chk(B.lib().omok_debug_set_children_kernel(self.h, int(which)))

    def set_profiling(self, on=True):
        """True / 1: time every launch; N > 1: time one search round in N (stats are scaled); False: off."""
        self._chk(B.lib().omok_set_profiling(self.h, int(on)))

    def stats(self):
        s = (C.c_double * len(B.STAT_NAMES))()
        self._chk(B.lib().omok_get_stats(self.h, s))
        return dict(zip(B.STAT_NAMES, list(s)))

    def reset_stats(self):
        self._chk(B.lib().omok_reset_stats(self.h))


class Environment:
    """environment::Environment (environment/src/lib.rs:62-166): the caller holds board / turn / legal_move_count, the
    device rules kernel applies place_stone to them (omok_env_place_stone, batch of one)."""

    def __init__(self, engine):
        self.eng = engine
        self._board = np.zeros((1, engine.hw), dtype=np.uint8)   # Environment::new (:73-79)
        self._turn = np.zeros(1, dtype=np.uint8)
        self._legal = np.full(1, engine.hw, dtype=np.uint16)

    board = property(lambda s: s._board[0])
    turn = property(lambda s: int(s._turn[0]))
    legal_move_count = property(lambda s: int(s._legal[0]))

    def place_stone(self, index):
        s = int(self.eng.env_place_stone(self._board, self._turn, self._legal, [int(index)])[0])
        return None if s < 0 else s

    def encode_board(self, turn):
        """Environment::encode_board(turn): the first 2*HW floats of the NN input with that perspective."""
        t = np.array([turn], dtype=np.uint8)
        return self.eng.encode_nn_input(self.board[None], t, B.MODE_PLAYER).reshape(-1)[: 2 * self.eng.hw].copy()


class SelfPlay:
    """G games x (black agent, white agent): the reference's self-play phase on one GPU."""

    def __init__(self, engine):
        self.eng = engine
        self.h = engine.h
        self.n, self.hw, self.games = engine.n, engine.hw, engine.games
        self._chk = engine._chk

    def reset(self):
        self._chk(B.lib().omok_selfplay_reset(self.h))

    def set_episode(self, episode):
        """index of the RNG stream the NEXT reset uses (each reset = one trainer iteration advances it by itself)"""
        self._chk(B.lib().omok_set_episode(self.h, int(episode)))

    def compute_policy(self):
        """Agent::compute_policy of the side-to-move agents: (pi [G][HW], has [G]); has == 0 where the reference returns None"""
        pi = np.zeros((self.games, self.hw), dtype=np.float32)
        has = np.zeros(self.games, dtype=np.uint8)
        self._chk(B.lib().omok_compute_policy(self.h, B.fptr(pi), B.u8ptr(has)))
        return pi, has

    def play_actions(self, actions):
        """externally chosen moves: ensure_action_exists + play_action on both agents of every live game"""
        a = np.ascontiguousarray(actions, dtype=np.int32)
        assert a.size == self.games
        self._chk(B.lib().omok_play_actions(self.h, B.iptr(a)))

    def set_actions(self, actions):
        """step-wise form of play_actions (mirror_generate / mirror_eval|inject / mirror_apply follow)"""
        a = np.ascontiguousarray(actions, dtype=np.int32)
        assert a.size == self.games
        self._chk(B.lib().omok_set_actions(self.h, B.iptr(a)))

    def root_children(self, game, side):
        """(actions, n, w, p) of the root's children in insertion order"""
        cap = self.hw
        a = np.zeros(cap, dtype=np.int32)
        n = np.zeros(cap, dtype=np.uint32)
        w = np.zeros(cap, dtype=np.float32)
        p = np.zeros(cap, dtype=np.float32)
        k = self._chk(B.lib().omok_root_children(self.h, game, side, B.iptr(a), n.ctypes.data_as(C.POINTER(C.c_uint32)), B.fptr(w), B.fptr(p), cap))
        return a[:k], n[:k], w[:k], p[:k]

    @property
    def ply(self):
        return self._chk(B.lib().omok_current_ply(self.h))

    @property
    def alive_count(self):
        return self._chk(B.lib().omok_alive_count(self.h))

    def game_info(self):
        alive = np.zeros(self.games, dtype=np.uint8)
        status = np.zeros(self.games, dtype=np.uint8)
        plies = np.zeros(self.games, dtype=np.int32)
        self._chk(B.lib().omok_game_info(self.h, B.u8ptr(alive), B.u8ptr(status), B.iptr(plies)))
        return alive, status, plies

    def execute(self, count, batch_size, epsilon=0.25, alpha=0.03):
        self._chk(B.lib().omok_execute(self.h, count, batch_size, epsilon, alpha))

    def execute_shared(self, count, batch_size, epsilon=0.25, alpha=0.03, waves=8):
        """MCTSExecutor::run: one tree (games = 1) searched by `waves` wavefronts"""
        self._chk(B.lib().omok_execute_shared(self.h, count, batch_size, epsilon, alpha, waves))

    def execute_shared_recorded(self, count, batch_size, epsilon=0.25, alpha=0.03, waves=8):
        """omok_execute_shared_recorded: returns a list of groups, each (sim_order [wave ids], backup_order [wave ids], p [req][HW], v [req])."""
        rounds = -(-count // batch_size)
        groups = -(-rounds // waves)
        per = waves * batch_size
        so = np.zeros((groups, per), dtype=np.uint8)
        bo = np.zeros((groups, per), dtype=np.uint8)
        gc = np.zeros((groups, 3), dtype=np.int32)
        cap = rounds * batch_size
        p = np.zeros((cap, self.hw), dtype=np.float32)
        v = np.zeros(cap, dtype=np.float32)
        ng, nr = C.c_int32(), C.c_int32()
        self._chk(B.lib().omok_execute_shared_recorded(self.h, count, batch_size, epsilon, alpha, waves, B.u8ptr(so), B.u8ptr(bo), B.iptr(gc), B.fptr(p), B.fptr(v),
                                                       cap, C.byref(ng), C.byref(nr)))
        out, base = [], 0
        for g in range(ng.value):
            ns, nb, nq = (int(x) for x in gc[g])
            out.append((so[g, :ns].copy(), bo[g, :nb].copy(), p[base:base + nq].copy(), v[base:base + nq].copy()))
            base += nq
        return out

    def sample_actions(self, temperature=1.0, threshold=30):
        a = np.zeros(self.games, dtype=np.int32)
        self._chk(B.lib().omok_sample_actions(self.h, temperature, threshold, B.iptr(a)))
        return a

    def advance(self):
        self._chk(B.lib().omok_advance(self.h))

    def run(self, count, batch_size, epsilon=0.25, alpha=0.03, temperature=1.0, threshold=30, max_plies=0):
        s = (C.c_double * len(B.STAT_NAMES))()
        self._chk(B.lib().omok_selfplay_run(self.h, count, batch_size, epsilon, alpha, temperature, threshold, max_plies, s))
        return dict(zip(B.STAT_NAMES, list(s)))

    def run_slots(self, total_games, count, batch_size, records_ptr, cap_records, epsilon=0.25, alpha=0.03, temperature=1.0, threshold=30):
        """Slots mode (omok_selfplay_run_slots): `total_games` games on the engine's slots, finished slots restarted with the next game
        index.  `records_ptr` = device buffer of cap_records x replay_record_bytes (e.g. a torch uint8 tensor's data_ptr()).
        Returns (stats, n_records, offsets[int64], lengths[int32], status[int32]) indexed by game index."""
        s = (C.c_double * len(B.STAT_NAMES))()
        off = np.zeros(total_games, dtype=np.int64)
        ln = np.zeros(total_games, dtype=np.int32)
        stt = np.zeros(total_games, dtype=np.int32)
        n = C.c_int64()
        self._chk(B.lib().omok_selfplay_run_slots(self.h, total_games, count, batch_size, epsilon, alpha, temperature, threshold,
                                                  C.c_void_p(records_ptr), cap_records, off.ctypes.data_as(C.POINTER(C.c_int64)),
                                                  ln.ctypes.data_as(C.POINTER(C.c_int32)), stt.ctypes.data_as(C.POINTER(C.c_int32)),
                                                  C.byref(n), s))
        return dict(zip(B.STAT_NAMES, list(s))), int(n.value), off, ln, stt

    # ---- step-wise (parity tests) -----------------------------------------------------------
    def round_generate(self, rnd, batch_size, epsilon=0.25, alpha=0.03):
        n = C.c_int32()
        self._chk(B.lib().omok_round_generate(self.h, rnd, batch_size, epsilon, alpha, C.byref(n)))
        self._nreq = n.value
        return n.value

    def round_inputs(self):
        out = np.zeros((self._nreq, 3 * self.hw), dtype=np.float32)
        if self._nreq:
            self._chk(B.lib().omok_round_inputs(self.h, B.fptr(out)))
        return out

    def round_eval(self):
        self._chk(B.lib().omok_round_eval(self.h))
        p = np.zeros((self._nreq, self.hw), dtype=np.float32)
        v = np.zeros(self._nreq, dtype=np.float32)
        if self._nreq:
            self._chk(B.lib().omok_round_outputs(self.h, B.fptr(p), B.fptr(v)))
        return p, v

    def round_logits(self):
        """pre-softmax policy logits [n][HW] and pre-tanh values [n] of the round just evaluated (call between round_eval and round_scatter)"""
        lg = np.zeros((self._nreq, self.hw), dtype=np.float32)
        vp = np.zeros(self._nreq, dtype=np.float32)
        if self._nreq:
            self._chk(B.lib().omok_round_logits(self.h, B.fptr(lg), B.fptr(vp)))
        return lg, vp

    def round_inject(self, p, v):
        p = np.ascontiguousarray(p, dtype=np.float32)
        v = np.ascontiguousarray(v, dtype=np.float32)
        if self._nreq:
            self._chk(B.lib().omok_round_inject(self.h, B.fptr(p), B.fptr(v)))

    def round_scatter(self):
        self._chk(B.lib().omok_round_scatter(self.h))

    def mirror_generate(self):
        n = C.c_int32()
        self._chk(B.lib().omok_mirror_generate(self.h, C.byref(n)))
        self._nmir = n.value
        return n.value

    def mirror_inputs(self):
        out = np.zeros((self._nmir, 3 * self.hw), dtype=np.float32)
        if self._nmir:
            self._chk(B.lib().omok_mirror_inputs(self.h, B.fptr(out)))
        return out

    def mirror_eval(self):
        self._chk(B.lib().omok_mirror_eval(self.h))
        p = np.zeros((self._nmir, self.hw), dtype=np.float32)
        if self._nmir:
            self._chk(B.lib().omok_mirror_outputs(self.h, B.fptr(p)))
        return p

    def mirror_inject(self, p):
        p = np.ascontiguousarray(p, dtype=np.float32)
        if self._nmir:
            self._chk(B.lib().omok_mirror_inject(self.h, B.fptr(p)))

    def mirror_apply(self):
        self._chk(B.lib().omok_mirror_apply(self.h))

    # ---- inspection -------------------------------------------------------------------------
    def tree_dump(self, game, side):
        cap = self.eng.max_nodes
        ints = np.zeros((cap, 8), dtype=np.int32)
        floats = np.zeros((cap, 1 + self.hw), dtype=np.float32)
        n = self._chk(B.lib().omok_tree_dump(self.h, game, side, B.iptr(ints), B.fptr(floats), cap))
        return ints[:n].copy(), floats[:n].copy()

    def tree_root(self, game, side):
        rn, rw, nn, nt = C.c_uint32(), C.c_float(), C.c_int32(), C.c_int32()
        self._chk(B.lib().omok_tree_root(self.h, game, side, C.byref(rn), C.byref(rw), C.byref(nn), C.byref(nt)))
        return rn.value, rw.value, nn.value, nt.value

    def replay(self, game):
        cap = self.hw
        boards = np.zeros((cap, self.hw), dtype=np.uint8)
        turns = np.zeros(cap, dtype=np.uint8)
        pi = np.zeros((cap, self.hw), dtype=np.float32)
        z = np.zeros(cap, dtype=np.float32)
        n = self._chk(B.lib().omok_replay_game(self.h, game, B.u8ptr(boards), B.u8ptr(turns), B.fptr(pi), B.fptr(z), cap))
        return boards[:n], turns[:n], pi[:n], z[:n]

    def replay_augmented(self, game):
        """Trainer::train replay post-processing for one game (src/trainer.rs:207-324): z back-fill, then the game's
        transitions followed by their 5 augmentations each (rot90, rot180, rot270, flipH, flipV)."""
        cap = 6 * self.hw
        boards = np.zeros((cap, self.hw), dtype=np.uint8)
        turns = np.zeros(cap, dtype=np.uint8)
        pi = np.zeros((cap, self.hw), dtype=np.float32)
        z = np.zeros(cap, dtype=np.float32)
        n = self._chk(B.lib().omok_replay_augmented_game(self.h, game, B.u8ptr(boards), B.u8ptr(turns), B.fptr(pi), B.fptr(z), cap))
        return boards[:n], turns[:n], pi[:n], z[:n]

    def replay_augment_into(self, dev_ptr, cap_records):
        n = B.lib().omok_replay_augment_dev(self.h, C.c_void_p(dev_ptr), cap_records)
        if n < 0:
            raise B.OmokError(int(n), B.lib().omok_last_error(self.h).decode())
        return int(n)

    def replay_record_bytes(self):
        return self._chk(B.lib().omok_replay_record_bytes(self.h))

    def replay_pack_into(self, dev_ptr, cap_records):
        n = B.lib().omok_replay_pack_dev(self.h, C.c_void_p(dev_ptr), cap_records)
        if n < 0:
            raise B.OmokError(int(n), B.lib().omok_last_error(self.h).decode())
        return int(n)
