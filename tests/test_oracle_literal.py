"""Oracle vs oracle: oracle/selfplay.c (the arena restatement the GPU tests compare the engine with) against
oracle/literal.c (the reference's own data structures: pointer nodes that store p, explicit placeholder policies and
refresh loops, recursive free, swap_remove of finished games).  The two share only the rules (pinned by the reference's
environment tests) and the build-defined RNG stream; identical canonical dumps, moves and replay tuples on the same (p, v)
rows mean the storage shortcuts of selfplay.c (and of the engine) are behaviour-preserving:
  - child.p is never stored      <->  literal child.p == parent.policy[action] at every comparison point
  - implicit placeholder policy  <->  literal explicit uniform rows
  - (n, w) in the parent's table, stable arena compaction  <->  literal per-node fields, recursive free, creation stamps
"""
import numpy as np
import pytest

import omok_ai_amd  # noqa: F401
from omok_ai_amd import weights
from oracle import oracle as O
from helpers import tree_shape


class FakeNet:
    """A cheap deterministic stand-in for the policy/value net: peaked softmax of a fixed random projection of the input
    row.  Both oracles get the SAME rows, so any function works; peaked policies make the search deep quickly."""

    def __init__(self, n, seed=0, scale=9.0):
        rng = np.random.default_rng(seed)
        hw = n * n
        self.w = rng.standard_normal((3 * hw, hw)).astype(np.float32)
        self.wv = rng.standard_normal(3 * hw).astype(np.float32)
        self.scale = scale

    def forward(self, x):
        x = np.ascontiguousarray(x, dtype=np.float32).reshape(len(x), -1)
        ones = 1.0 + x.sum(axis=1, keepdims=True)
        logits = (x @ self.w) * (self.scale / np.sqrt(ones))
        logits -= logits.max(axis=1, keepdims=True)
        e = np.exp(logits)
        p = (e / e.sum(axis=1, keepdims=True)).astype(np.float32)
        v = np.tanh((x @ self.wv) / np.sqrt(ones[:, 0])).astype(np.float32)
        return p, v


class RealNet:
    def __init__(self, n):
        self.net = O.Net(n, weights.init_random(n, seed=0))

    def forward(self, x):
        return self.net.forward(x, threads=8)


def _perm(games_a, games_l):
    """row index into A's batch for every row of L's batch: the j-th row of game g on both sides"""
    where = {}
    for r, g in enumerate(games_a):
        where.setdefault(int(g), []).append(r)
    seen = {}
    out = []
    for g in games_l:
        j = seen.get(int(g), 0)
        out.append(where[int(g)][j])
        seen[int(g)] = j + 1
    return np.array(out, dtype=np.int64)


def _compare(A, L, games, tag):
    total_nodes = 0
    for g in range(games):
        assert A.game_alive(g) == L.game_alive(g), f"{tag}: game {g} alive"
        if not A.game_alive(g):
            continue  # the reference drops the agents of a finished game (swap_remove)
        for side in (0, 1):
            ai, af = A.tree_dump(g, side)
            li, lf = L.tree_dump(g, side)
            assert ai.shape == li.shape, f"{tag}: game {g} side {side}: {len(ai)} vs {len(li)} nodes"
            assert np.array_equal(ai[:, :7], li[:, :7]), f"{tag}: node records (game {g} side {side})"
            assert np.array_equal(ai[:, 7] & 0xFFFF, li[:, 7] & 0xFFFF), f"{tag}: insertion ranks (game {g} side {side})"
            assert np.array_equal(af.view(np.uint32), lf.view(np.uint32)), f"{tag}: w / policy bits (game {g} side {side})"
            p_stored, p_parent = L.tree_priors(g, side)
            assert np.array_equal(p_stored.view(np.uint32), p_parent.view(np.uint32)), f"{tag}: child.p != parent.policy[action]"
            total_nodes += len(ai)
    assert L.live_nodes == total_nodes, f"{tag}: allocator balance {L.live_nodes} vs {total_nodes}"


def _drive(n, games, count, k, max_plies, net, threshold=30, seed=7, offset=5, episode=None, external_every=0, ext_seed=0):
    hw = n * n
    root_p, _ = net.forward(O.Environment(n).encode_nn_input(0)[None])
    cap = min(16384, 4 * count + 1024)
    A = O.SelfPlay(n, games, cap_nodes=cap, cap_tables=max(256, cap // 4), seed=seed, game_offset=offset)
    L = O.Literal(n, games, seed=seed, game_offset=offset, cap_nodes=cap)
    if episode is not None:
        A.set_episode(episode)
        L.set_episode(episode)
    A.reset(root_p[0])
    L.reset(root_p[0])
    _compare(A, L, games, "reset")
    rng = np.random.default_rng(ext_seed)
    played = [set() for _ in range(games)]
    shape = [0, 0, 0]
    ply = 0
    while A.alive_count > 0 and (max_plies == 0 or ply < max_plies):
        for rnd in range((count + k - 1) // k):
            in_a = A.round_generate(rnd, k, 0.25, 0.03)
            in_l, g_l = L.round_generate(rnd, k, 0.25, 0.03)
            assert len(in_a) == len(in_l), f"ply {ply} round {rnd}: request count"
            if len(in_a) == 0:
                continue
            g_a = [A.request_info(r)[0] for r in range(len(in_a))]
            perm = _perm(g_a, g_l)
            assert np.array_equal(in_l, in_a[perm]), f"ply {ply} round {rnd}: request boards"
            if rnd == 0:
                _compare(A, L, games, f"ply {ply} round 0 generated")
            p, v = net.forward(in_a)
            A.round_scatter(p, v)
            L.round_scatter(p[perm], v[perm])
        _compare(A, L, games, f"ply {ply} after execute")
        for g in range(games):
            if A.game_alive(g):
                full, full_nr, depth = tree_shape(A.tree_dump(g, ply & 1)[0])
                shape = [max(shape[0], full), max(shape[1], full_nr), max(shape[2], depth)]
                pa, pl = A.compute_policy(g), L.compute_policy(g)
                assert (pa is None) == (pl is None) and (pa is None or np.array_equal(pa.view(np.uint32), pl.view(np.uint32)))
        external = external_every > 0 and ply % external_every == external_every - 1
        if external:  # externally chosen legal moves, most of them NOT in the tree
            acts = np.full(games, -1, dtype=np.int32)
            for g in range(games):
                if A.game_alive(g):
                    empties = [c for c in range(hw) if c not in played[g]]
                    assert len(empties) == int(A.tree_dump(g, ply & 1)[0][0, 4])  # the root's legal_move_count
                    acts[g] = int(rng.choice(empties))
            A.set_actions(acts)
            L.set_actions(acts)
        else:
            acts = A.sample(1.0, threshold)
            assert np.array_equal(acts, L.sample(1.0, threshold)), f"ply {ply}: actions"
        for g in range(games):
            if acts[g] >= 0:
                played[g].add(int(acts[g]))
        m_a = A.mirror_generate()
        m_l, mg_l = L.mirror_generate()
        alive_games = [g for g in range(games) if A.game_alive(g)]
        perm = _perm(alive_games, mg_l)
        assert np.array_equal(m_l, m_a[perm]), f"ply {ply}: mirror inputs"
        pm, _ = net.forward(m_a)
        A.advance(pm)
        L.advance(pm[perm], external=external)
        assert A.error == 0 and L.error == 0, (A.error, L.error)
        _compare(A, L, games, f"ply {ply} after advance")
        for g in range(games):
            assert A.game_status(g) == L.game_status(g) and A.game_plies(g) == L.game_plies(g)
        ply += 1
    for g in range(games):
        ab, at, ap, az = A.replay(g)
        lb, lt, lp, lz = L.replay(g)
        assert np.array_equal(ab, lb) and np.array_equal(at, lt) and np.array_equal(az, lz)
        assert np.array_equal(ap.view(np.uint32), lp.view(np.uint32))
    return tuple(shape), ply


@pytest.mark.parametrize("n,games,count,k,max_plies,min_depth", [
    (9, 5, 48, 8, 0, 1),       # whole games, swap_remove order diverges from game order as games finish
    (9, 3, 200, 8, 0, 3),      # configs[2] per-tree workload: deep trees, fully expanded nodes
    (15, 2, 800, 16, 3, 3),    # configs[1]: the benchmark regime
    (15, 1, 100, 16, 12, 1),   # configs[0]: count not a multiple of K
])
def test_literal_equals_arena_oracle_fake_net(n, games, count, k, max_plies, min_depth):
    shape, plies = _drive(n, games, count, k, max_plies, FakeNet(n))
    print(f"shape (full, full non-root, depth) = {shape}, {plies} plies")
    assert shape[2] >= min_depth


def test_literal_equals_arena_oracle_real_net_and_episode_stream():
    """the random-init net of the GPU tests (flat policies: the wide-and-shallow regime), on a non-zero episode stream"""
    _drive(9, 3, 40, 16, 10, RealNet(9), threshold=4, episode=3)


def test_external_moves_on_both_oracles():
    """gui / benchmark style play: every third ply the move comes from outside (mostly a cell the search never expanded):
    ensure_action_exists + play_action on BOTH agents (agent.rs:144-232), no transition recorded."""
    shape, plies = _drive(9, 4, 32, 8, 24, FakeNet(9, seed=2), external_every=3, ext_seed=1)
    assert plies >= 6


@pytest.mark.parametrize("n,count,k", [(9, 64, 8), (15, 48, 16)])
def test_shared_run_with_one_task_at_a_time_is_the_round_loop(n, count, k):
    """MCTSExecutor::run with the tasks run one after the other (waves = 1: every group is one round, its K simulations in order, its
    requests scattered in order) is ParallelMCTSExecutor::execute on one agent: the two entry points of oracle/literal.c must leave
    identical trees (they share one_simulation / propagate; this pins the scheduling code around them)."""
    net = FakeNet(n, seed=2)
    root_p = net.forward(O.Environment(n).encode_nn_input(0)[None])[0][0]
    a = O.Literal(n, 1, seed=9)
    b = O.Literal(n, 1, seed=9)
    a.reset(root_p)
    b.reset(root_p)
    lib, u8p = O.lib(), O.C.POINTER(O.C.c_uint8)
    for ply in range(5):
        rounds = -(-count // k)
        for rnd in range(rounds):
            x, _ = a.round_generate(rnd, k, 0.25, 0.03)
            p, v = net.forward(x) if len(x) else (np.zeros((0, n * n), np.float32), np.zeros(0, np.float32))
            a.round_scatter(p, v)
        lib.lit_shared_noise(b.h, 0.25, 0.03)
        for g in range(rounds):
            so = np.zeros(k, dtype=np.uint8)
            inp = np.zeros((k, 3 * n * n), dtype=np.float32)
            m = lib.lit_shared_group_generate(b.h, g, 1, rounds, k, so.ctypes.data_as(u8p), k, O._fp(inp), k)
            assert m >= 0 and b.error == 0
            p, v = net.forward(inp[:m]) if m else (np.zeros((0, n * n), np.float32), np.zeros(0, np.float32))
            bo = np.zeros(m, dtype=np.uint8)
            lib.lit_shared_group_scatter(b.h, O._fp(np.ascontiguousarray(p, dtype=np.float32)), O._fp(np.ascontiguousarray(v, dtype=np.float32)),
                                         bo.ctypes.data_as(u8p), m)
            assert b.error == 0
        side = ply & 1
        ai, af = a.tree_dump(0, side)
        bi, bf = b.tree_dump(0, side)
        assert np.array_equal(ai, bi) and np.array_equal(af.view(np.uint32), bf.view(np.uint32)), f"ply {ply}"
        act = a.sample(1.0, 3)
        assert np.array_equal(act, b.sample(1.0, 3))
        xa, _ = a.mirror_generate()
        xb, _ = b.mirror_generate()
        assert np.array_equal(xa, xb)
        pm = net.forward(xa)[0] if len(xa) else np.zeros((0, n * n), np.float32)
        a.advance(pm)
        b.advance(pm)
        if a.alive_count == 0:
            break
    assert a.live_nodes == b.live_nodes


def test_shared_run_rejects_a_bad_schedule():
    n, k = 9, 4
    net = FakeNet(n, seed=1)
    lit = O.Literal(n, 1, seed=1)
    lit.reset(net.forward(O.Environment(n).encode_nn_input(0)[None])[0][0])
    lib, u8p = O.lib(), O.C.POINTER(O.C.c_uint8)
    so = np.array([0, 1, 0, 1, 0, 1, 0], dtype=np.uint8)  # task 1 runs 3 of its 4 simulations
    assert lib.lit_shared_group_generate(lit.h, 0, 2, 2, k, so.ctypes.data_as(u8p), len(so), None, 8) == -1 and lit.error == 4
