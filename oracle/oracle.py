"""ctypes loader for the CPU oracle (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package (omok-ai_amd/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libomok_oracle.so")

MAX_HW = 225
EMPTY, BLACK, WHITE = 0, 1, 2
TURN_BLACK, TURN_WHITE = 0, 1
IN_PROGRESS, DRAW, BLACK_WIN, WHITE_WIN = 0, 1, 2, 3
MODE_PLAYER, MODE_OPPONENT = 0, 1


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


class Env(C.Structure):
    _fields_ = [("n", C.c_int32), ("turn", C.c_uint8), ("legal", C.c_uint16), ("board", C.c_uint8 * MAX_HW)]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB):
        build()
    L = C.CDLL(_LIB)
    fp = C.POINTER(C.c_float)
    L.orc_env_init.argtypes = [C.POINTER(Env), C.c_int]
    L.orc_env_place_stone.argtypes = [C.POINTER(Env), C.c_int]
    L.orc_env_place_stone.restype = C.c_int
    L.orc_env_encode_board.argtypes = [C.POINTER(Env), C.c_int, fp]
    L.orc_encode_nn_input.argtypes = [C.POINTER(Env), C.c_int, fp]
    for name in ("orc_rotate_90", "orc_rotate_180", "orc_rotate_270", "orc_flip_horizontal", "orc_flip_vertical"):
        getattr(L, name).argtypes = [fp, fp, C.c_int]
    u8 = C.POINTER(C.c_uint8)
    L.orc_replay_postprocess.argtypes = [C.c_int, C.c_int, u8, u8, fp, fp, u8, u8, fp, fp]
    L.orc_replay_postprocess.restype = None
    L.orc_philox.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]
    L.orc_det_log.argtypes = [C.c_double]
    L.orc_det_log.restype = C.c_double
    L.orc_det_exp.argtypes = [C.c_double]
    L.orc_det_exp.restype = C.c_double
    L.orc_det_expf.argtypes = [C.c_float]
    L.orc_det_expf.restype = C.c_float
    L.orc_gamma.argtypes = [C.c_float, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32]
    L.orc_gamma.restype = C.c_float
    L.orc_net_create.argtypes = [C.c_int]
    L.orc_net_create.restype = C.c_void_p
    L.orc_net_destroy.argtypes = [C.c_void_p]
    L.orc_net_num_tensors.restype = C.c_int
    L.orc_net_tensor_size.argtypes = [C.c_void_p, C.c_int]
    L.orc_net_tensor_size.restype = C.c_int64
    L.orc_net_load.argtypes = [C.c_void_p, C.c_int, fp, C.c_int64]
    L.orc_net_load.restype = C.c_int
    L.orc_net_forward.argtypes = [C.c_void_p, fp, C.c_int, fp, fp, C.c_int]
    L.orc_net_forward_logits.argtypes = [C.c_void_p, fp, C.c_int, fp, fp, fp, fp, C.c_int]
    L.orc_sp_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_int64]
    L.orc_sp_create.restype = C.c_void_p
    L.orc_sp_destroy.argtypes = [C.c_void_p]
    L.orc_sp_reset.argtypes = [C.c_void_p, fp]
    L.orc_sp_set_episode.argtypes = [C.c_void_p, C.c_uint64]
    L.orc_sp_set_threads.argtypes = [C.c_void_p, C.c_int]
    L.orc_sp_set_actions.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
    L.orc_sp_compute_policy.argtypes = [C.c_void_p, C.c_int, fp]
    L.orc_sp_compute_policy.restype = C.c_int
    L.orc_stream_key.argtypes = [C.c_uint64, C.c_uint64]
    L.orc_stream_key.restype = C.c_uint64
    # oracle No. 2 (literal.c)
    i32p = C.POINTER(C.c_int32)
    L.lit_create.argtypes = [C.c_int, C.c_int, C.c_uint64, C.c_int64]
    L.lit_create.restype = C.c_void_p
    L.lit_destroy.argtypes = [C.c_void_p]
    L.lit_set_episode.argtypes = [C.c_void_p, C.c_uint64]
    L.lit_reset.argtypes = [C.c_void_p, fp]
    for name in ("lit_ply", "lit_error", "lit_alive_count"):
        getattr(L, name).argtypes = [C.c_void_p]
        getattr(L, name).restype = C.c_int
    L.lit_live_nodes.argtypes = [C.c_void_p]
    L.lit_live_nodes.restype = C.c_long
    for name in ("lit_game_alive", "lit_game_status", "lit_game_plies"):
        getattr(L, name).argtypes = [C.c_void_p, C.c_int]
        getattr(L, name).restype = C.c_int
    L.lit_round_generate.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, fp, i32p, C.c_int]
    L.lit_round_generate.restype = C.c_int
    L.lit_round_scatter.argtypes = [C.c_void_p, fp, fp]
    L.lit_sample.argtypes = [C.c_void_p, C.c_float, C.c_int, i32p]
    L.lit_shared_noise.argtypes = [C.c_void_p, C.c_float, C.c_float]
    L.lit_shared_group_generate.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint8), C.c_int, fp, C.c_int]
    L.lit_shared_group_scatter.argtypes = [C.c_void_p, fp, fp, C.POINTER(C.c_uint8), C.c_int]
    L.lit_set_actions.argtypes = [C.c_void_p, i32p]
    L.lit_mirror_generate.argtypes = [C.c_void_p, fp, i32p, C.c_int]
    L.lit_mirror_generate.restype = C.c_int
    L.lit_advance.argtypes = [C.c_void_p, fp, C.c_int]
    L.lit_compute_policy.argtypes = [C.c_void_p, C.c_int, fp]
    L.lit_compute_policy.restype = C.c_int
    L.lit_tree_dump.argtypes = [C.c_void_p, C.c_int, C.c_int, i32p, fp, C.c_int]
    L.lit_tree_dump.restype = C.c_int
    L.lit_tree_priors.argtypes = [C.c_void_p, C.c_int, C.c_int, fp, fp, C.c_int]
    L.lit_tree_priors.restype = C.c_int
    L.lit_replay.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint8), C.POINTER(C.c_uint8), fp, fp, C.c_int]
    L.lit_replay.restype = C.c_int
    for name in ("orc_sp_ply", "orc_sp_alive_count", "orc_sp_error"):
        getattr(L, name).argtypes = [C.c_void_p]
        getattr(L, name).restype = C.c_int
    for name in ("orc_sp_game_alive", "orc_sp_game_status", "orc_sp_game_plies"):
        getattr(L, name).argtypes = [C.c_void_p, C.c_int]
        getattr(L, name).restype = C.c_int
    L.orc_sp_round_generate.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, fp, C.c_int]
    L.orc_sp_round_generate.restype = C.c_int
    L.orc_sp_round_scatter.argtypes = [C.c_void_p, fp, fp]
    L.orc_sp_request_info.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.orc_sp_sample.argtypes = [C.c_void_p, C.c_float, C.c_int, C.POINTER(C.c_int32)]
    L.orc_sp_mirror_generate.argtypes = [C.c_void_p, fp, C.c_int]
    L.orc_sp_mirror_generate.restype = C.c_int
    L.orc_sp_advance.argtypes = [C.c_void_p, fp]
    L.orc_sp_tree_dump.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int32), fp, C.c_int]
    L.orc_sp_tree_dump.restype = C.c_int
    L.orc_sp_tree_root.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_uint32), fp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.orc_sp_replay.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint8), C.POINTER(C.c_uint8), fp, fp, C.c_int]
    L.orc_sp_replay.restype = C.c_int
    L.orc_selfplay_run.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float,
                                   C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
    L.orc_selfplay_run.restype = C.c_int
    _lib = L
    return L


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class Environment:
    """Mirror of environment::Environment (environment/src/lib.rs:62-166)."""

    def __init__(self, n=9):
        self.e = Env()
        lib().orc_env_init(C.byref(self.e), n)
        self.n = n

    @property
    def turn(self):
        return self.e.turn

    @property
    def legal_move_count(self):
        return self.e.legal

    @property
    def board(self):
        return np.array(self.e.board[: self.n * self.n], dtype=np.uint8)

    def place_stone(self, index):
        s = lib().orc_env_place_stone(C.byref(self.e), int(index))
        return None if s < 0 else s

    def encode_board(self, turn):
        out = np.zeros(2 * self.n * self.n, dtype=np.float32)
        lib().orc_env_encode_board(C.byref(self.e), int(turn), _fp(out))
        return out

    def encode_nn_input(self, mode=MODE_PLAYER):
        out = np.zeros(3 * self.n * self.n, dtype=np.float32)
        lib().orc_encode_nn_input(C.byref(self.e), int(mode), _fp(out))
        return out


class Net:
    def __init__(self, n, tensors=None):
        self.n = n
        self.h = lib().orc_net_create(n)
        if tensors is not None:
            self.load(tensors)

    def load(self, tensors):
        assert len(tensors) == 31
        for i, t in enumerate(tensors):
            t = np.ascontiguousarray(t, dtype=np.float32).ravel()
            assert lib().orc_net_load(self.h, i, _fp(t), t.size) == 0, f"tensor {i} has wrong size {t.size}"

    def forward(self, inputs, threads=1):
        inputs = np.ascontiguousarray(inputs, dtype=np.float32).reshape(-1, 3 * self.n * self.n)
        b = inputs.shape[0]
        p = np.zeros((b, self.n * self.n), dtype=np.float32)
        v = np.zeros(b, dtype=np.float32)
        lib().orc_net_forward(self.h, _fp(inputs), b, _fp(p), _fp(v), threads)
        return p, v

    def forward_logits(self, inputs, threads=1):
        """(p, v, logits in front of the softmax, value in front of tanh)"""
        inputs = np.ascontiguousarray(inputs, dtype=np.float32).reshape(-1, 3 * self.n * self.n)
        b = inputs.shape[0]
        p = np.zeros((b, self.n * self.n), dtype=np.float32)
        lg = np.zeros((b, self.n * self.n), dtype=np.float32)
        v = np.zeros(b, dtype=np.float32)
        vp = np.zeros(b, dtype=np.float32)
        lib().orc_net_forward_logits(self.h, _fp(inputs), b, _fp(p), _fp(v), _fp(lg), _fp(vp), threads)
        return p, v, lg, vp

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_net_destroy(self.h)
            self.h = None


class SelfPlay:
    """G games x two trees; step-wise mirror of trainer.rs:95-205 / pme.rs:26-270."""

    def __init__(self, n, games, cap_nodes=4096, cap_tables=2048, seed=0, game_offset=0):
        self.n, self.hw, self.games = n, n * n, games
        self.cap_nodes = cap_nodes
        self.h = lib().orc_sp_create(n, games, cap_nodes, cap_tables, seed, game_offset)
        assert self.h

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_sp_destroy(self.h)
            self.h = None

    def reset(self, root_policy):
        rp = np.ascontiguousarray(root_policy, dtype=np.float32).ravel()
        assert rp.size == self.hw
        lib().orc_sp_reset(self.h, _fp(rp))

    def set_episode(self, episode):
        """index of the RNG stream the NEXT reset uses (every reset = one trainer iteration advances it)"""
        lib().orc_sp_set_episode(self.h, int(episode))

    def set_threads(self, threads):
        """threads > 1: the per-game loops of round_generate / round_scatter run under OpenMP (same results)"""
        lib().orc_sp_set_threads(self.h, int(threads))

    def set_actions(self, actions):
        """externally chosen moves for every alive game: replaces sample(); mirror_generate() / advance() follow"""
        a = np.ascontiguousarray(actions, dtype=np.int32)
        assert a.size == self.games
        lib().orc_sp_set_actions(self.h, a.ctypes.data_as(C.POINTER(C.c_int32)))

    def compute_policy(self, game):
        """Agent::compute_policy of the side-to-move agent (agent.rs:43-77); None like the reference"""
        pol = np.zeros(self.hw, dtype=np.float32)
        return pol if lib().orc_sp_compute_policy(self.h, game, _fp(pol)) else None

    ply = property(lambda s: lib().orc_sp_ply(s.h))
    alive_count = property(lambda s: lib().orc_sp_alive_count(s.h))
    error = property(lambda s: lib().orc_sp_error(s.h))

    def game_alive(self, g):
        return lib().orc_sp_game_alive(self.h, g)

    def game_status(self, g):
        return lib().orc_sp_game_status(self.h, g)

    def game_plies(self, g):
        return lib().orc_sp_game_plies(self.h, g)

    def round_generate(self, rnd, batch_size, epsilon, alpha):
        buf = np.zeros((self.games * batch_size, 3 * self.hw), dtype=np.float32)
        b = lib().orc_sp_round_generate(self.h, rnd, batch_size, epsilon, alpha, _fp(buf), buf.shape[0])
        assert b >= 0
        return buf[:b]

    def request_info(self, r):
        g, nd = C.c_int(), C.c_int()
        lib().orc_sp_request_info(self.h, r, C.byref(g), C.byref(nd))
        return g.value, nd.value

    def round_scatter(self, p, v):
        p = np.ascontiguousarray(p, dtype=np.float32)
        v = np.ascontiguousarray(v, dtype=np.float32)
        lib().orc_sp_round_scatter(self.h, _fp(p), _fp(v))

    def sample(self, temperature, threshold):
        a = np.zeros(self.games, dtype=np.int32)
        lib().orc_sp_sample(self.h, temperature, threshold, a.ctypes.data_as(C.POINTER(C.c_int32)))
        return a

    def mirror_generate(self):
        buf = np.zeros((self.games, 3 * self.hw), dtype=np.float32)
        m = lib().orc_sp_mirror_generate(self.h, _fp(buf), self.games)
        assert m >= 0
        return buf[:m]

    def advance(self, p):
        p = np.ascontiguousarray(p, dtype=np.float32)
        lib().orc_sp_advance(self.h, _fp(p))

    def tree_dump(self, game, side):
        ints = np.zeros((self.cap_nodes, 8), dtype=np.int32)
        floats = np.zeros((self.cap_nodes, 1 + self.hw), dtype=np.float32)
        n = lib().orc_sp_tree_dump(self.h, game, side, ints.ctypes.data_as(C.POINTER(C.c_int32)), _fp(floats), self.cap_nodes)
        assert n >= 0
        return ints[:n].copy(), floats[:n].copy()

    def tree_root(self, game, side):
        rn, rw, nn, nt = C.c_uint32(), C.c_float(), C.c_int(), C.c_int()
        lib().orc_sp_tree_root(self.h, game, side, C.byref(rn), C.byref(rw), C.byref(nn), C.byref(nt))
        return rn.value, rw.value, nn.value, nt.value

    def replay(self, game):
        cap = self.hw
        boards = np.zeros((cap, self.hw), dtype=np.uint8)
        turns = np.zeros(cap, dtype=np.uint8)
        pi = np.zeros((cap, self.hw), dtype=np.float32)
        z = np.zeros(cap, dtype=np.float32)
        n = lib().orc_sp_replay(self.h, game, boards.ctypes.data_as(C.POINTER(C.c_uint8)),
                                turns.ctypes.data_as(C.POINTER(C.c_uint8)), _fp(pi), _fp(z), cap)
        return boards[:n], turns[:n], pi[:n], z[:n]

    def run(self, net, count, batch_size, epsilon=0.25, alpha=0.03, temperature=1.0, threshold=30,
            max_plies=0, threads=1):
        stats = (C.c_double * 6)()
        err = lib().orc_selfplay_run(self.h, net.h, count, batch_size, epsilon, alpha, temperature, threshold,
                                     max_plies, threads, stats)
        keys = ("sims", "evals", "ply_games", "finished", "t_net", "t_total")
        return err, dict(zip(keys, list(stats)))


class Literal:
    """Oracle No. 2 (oracle/literal.c): the reference's own data structures (pointer nodes storing p, explicit placeholder
    policies and refresh loops, recursive free, swap_remove of finished games).  Same driving surface as SelfPlay, but
    requests / mirror rows come in the reference's SLOT order: round_generate / mirror_generate also return the game id
    of every row, and round_scatter / advance take rows in that same order."""

    def __init__(self, n, games, seed=0, game_offset=0, cap_nodes=4096):
        self.n, self.hw, self.games = n, n * n, games
        self.cap_nodes = cap_nodes  # dump buffer size only: the literal tree has no arena
        self.h = lib().lit_create(n, games, seed, game_offset)
        assert self.h

    def __del__(self):
        if getattr(self, "h", None):
            lib().lit_destroy(self.h)
            self.h = None

    def set_episode(self, episode):
        lib().lit_set_episode(self.h, int(episode))

    def reset(self, root_policy):
        rp = np.ascontiguousarray(root_policy, dtype=np.float32).ravel()
        assert rp.size == self.hw
        lib().lit_reset(self.h, _fp(rp))

    ply = property(lambda s: lib().lit_ply(s.h))
    alive_count = property(lambda s: lib().lit_alive_count(s.h))
    error = property(lambda s: lib().lit_error(s.h))
    live_nodes = property(lambda s: lib().lit_live_nodes(s.h))

    def game_alive(self, g):
        return lib().lit_game_alive(self.h, g)

    def game_status(self, g):
        return lib().lit_game_status(self.h, g)

    def game_plies(self, g):
        return lib().lit_game_plies(self.h, g)

    def round_generate(self, rnd, batch_size, epsilon, alpha):
        buf = np.zeros((self.games * batch_size, 3 * self.hw), dtype=np.float32)
        games = np.zeros(self.games * batch_size, dtype=np.int32)
        b = lib().lit_round_generate(self.h, rnd, batch_size, epsilon, alpha, _fp(buf), games.ctypes.data_as(C.POINTER(C.c_int32)), buf.shape[0])
        assert b >= 0
        return buf[:b], games[:b]

    def round_scatter(self, p, v):
        p = np.ascontiguousarray(p, dtype=np.float32)
        v = np.ascontiguousarray(v, dtype=np.float32)
        lib().lit_round_scatter(self.h, _fp(p), _fp(v))

    def shared_run(self, count, batch_size, epsilon, alpha, waves, groups):
        """MCTSExecutor::run on game 0 under a recorded interleaving: `groups` = [(sim_order, backup_order, p, v), ...] as returned by the
        engine's execute_shared_recorded.  Returns the request inputs of every group (for comparison with the engine's)."""
        rounds = -(-count // batch_size)
        lib().lit_shared_noise(self.h, epsilon, alpha)
        inputs = []
        u8p = C.POINTER(C.c_uint8)
        for g, (so, bo, p, v) in enumerate(groups):
            so = np.ascontiguousarray(so, dtype=np.uint8)
            bo = np.ascontiguousarray(bo, dtype=np.uint8)
            cap = waves * batch_size
            inp = np.zeros((cap, 3 * self.hw), dtype=np.float32)
            n = lib().lit_shared_group_generate(self.h, g, waves, rounds, batch_size, so.ctypes.data_as(u8p), len(so), _fp(inp), cap)
            assert n >= 0 and self.error == 0, f"group {g}: schedule rejected (error {self.error})"
            assert n == len(v) == len(bo), f"group {g}: {n} requests, the recording has {len(v)} outputs / {len(bo)} backups"
            inputs.append(inp[:n].copy())
            p = np.ascontiguousarray(p, dtype=np.float32).reshape(n, self.hw) if n else np.zeros((0, self.hw), dtype=np.float32)
            v = np.ascontiguousarray(v, dtype=np.float32)
            lib().lit_shared_group_scatter(self.h, _fp(p), _fp(v), bo.ctypes.data_as(u8p), len(bo))
            assert self.error == 0, f"group {g}: backup order rejected"
        return inputs

    def sample(self, temperature, threshold):
        a = np.zeros(self.games, dtype=np.int32)
        lib().lit_sample(self.h, temperature, threshold, a.ctypes.data_as(C.POINTER(C.c_int32)))
        return a

    def set_actions(self, actions):
        a = np.ascontiguousarray(actions, dtype=np.int32)
        assert a.size == self.games
        lib().lit_set_actions(self.h, a.ctypes.data_as(C.POINTER(C.c_int32)))

    def mirror_generate(self):
        buf = np.zeros((self.games, 3 * self.hw), dtype=np.float32)
        games = np.zeros(self.games, dtype=np.int32)
        m = lib().lit_mirror_generate(self.h, _fp(buf), games.ctypes.data_as(C.POINTER(C.c_int32)), self.games)
        assert m >= 0
        return buf[:m], games[:m]

    def advance(self, p, external=False):
        p = np.ascontiguousarray(p, dtype=np.float32)
        lib().lit_advance(self.h, _fp(p), 1 if external else 0)

    def compute_policy(self, game):
        pol = np.zeros(self.hw, dtype=np.float32)
        return pol if lib().lit_compute_policy(self.h, game, _fp(pol)) else None

    def tree_dump(self, game, side):
        ints = np.zeros((self.cap_nodes, 8), dtype=np.int32)
        floats = np.zeros((self.cap_nodes, 1 + self.hw), dtype=np.float32)
        n = lib().lit_tree_dump(self.h, game, side, ints.ctypes.data_as(C.POINTER(C.c_int32)), _fp(floats), self.cap_nodes)
        assert n >= 0, "raise cap_nodes"
        return ints[:n].copy(), floats[:n].copy()

    def tree_priors(self, game, side):
        """(child.p as stored, parent.policy[child.action]) per node in dump order: the reference keeps them equal"""
        a = np.zeros(self.cap_nodes, dtype=np.float32)
        b = np.zeros(self.cap_nodes, dtype=np.float32)
        n = lib().lit_tree_priors(self.h, game, side, _fp(a), _fp(b), self.cap_nodes)
        assert n >= 0
        return a[:n].copy(), b[:n].copy()

    def replay(self, game):
        cap = self.hw
        boards = np.zeros((cap, self.hw), dtype=np.uint8)
        turns = np.zeros(cap, dtype=np.uint8)
        pi = np.zeros((cap, self.hw), dtype=np.float32)
        z = np.zeros(cap, dtype=np.float32)
        n = lib().lit_replay(self.h, game, boards.ctypes.data_as(C.POINTER(C.c_uint8)), turns.ctypes.data_as(C.POINTER(C.c_uint8)), _fp(pi), _fp(z), cap)
        return boards[:n], turns[:n], pi[:n], z[:n]


def stream_key(seed, episode):
    return int(lib().orc_stream_key(int(seed), int(episode)))


def replay_postprocess(n, boards, turns, pi, z):
    """Trainer::train replay post-processing of one game (src/trainer.rs:207-324): returns 6*len records."""
    hw = n * n
    ln = len(turns)
    boards = np.ascontiguousarray(boards, dtype=np.uint8).reshape(ln, hw)
    turns = np.ascontiguousarray(turns, dtype=np.uint8)
    pi = np.ascontiguousarray(pi, dtype=np.float32).reshape(ln, hw)
    z = np.ascontiguousarray(z, dtype=np.float32)
    bo = np.zeros((6 * ln, hw), dtype=np.uint8)
    to = np.zeros(6 * ln, dtype=np.uint8)
    po = np.zeros((6 * ln, hw), dtype=np.float32)
    zo = np.zeros(6 * ln, dtype=np.float32)
    u8 = C.POINTER(C.c_uint8)
    fp = C.POINTER(C.c_float)
    lib().orc_replay_postprocess(n, ln, boards.ctypes.data_as(u8), turns.ctypes.data_as(u8), pi.ctypes.data_as(fp), z.ctypes.data_as(fp),
                                 bo.ctypes.data_as(u8), to.ctypes.data_as(u8), po.ctypes.data_as(fp), zo.ctypes.data_as(fp))
    return bo, to, po, zo
