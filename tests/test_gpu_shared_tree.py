"""One tree searched by many wavefronts (omok_execute_shared = MCTSExecutor::run, alpha-zero/src/mcts_executor.rs:29-255;
SURVEY 8f rank 4).  The reference's version is nondeterministic by design (rayon tasks racing on one tree), so parity is:
  - waves = 1 is the sequential schedule: identical to omok_execute bit for bit, ply after ply;
  - waves >= 1 under a RECORDED interleaving (omok_execute_shared_recorded: every simulation / backup under the tree lock, the lock
    order written down): the literal oracle replays exactly that schedule (oracle/literal.c lit_shared_*) and the trees, moves and
    request boards are identical bit for bit, ply after ply -- an independent check of MCTSExecutor::run's arithmetic
    (mcts_executor.rs:76-255: shared n / w, expansion, terminal shortcuts, batch evaluation, scatter) for W = 1, 8 and 16;
  - waves > 1, free-running: every structural invariant of the tree holds, visit counts are conserved (node n >= sum of its children's n),
    nothing is lost or duplicated in the arena, every non-terminal node got its evaluation, and the search still concentrates
    on the same moves as the sequential one (loose statistical bound, printed)."""
import numpy as np
import pytest

import omok_ai_amd as oa
from oracle import oracle as O
from test_oracle_selfplay import check_tree_invariants

pytestmark = pytest.mark.gpu


def _engine(n, k, seed, waves, max_nodes=4096, net_mode=0):
    eng = oa.Engine(board_size=n, games=1, max_nodes=max_nodes, max_tables=max_nodes // 2, max_batch_k=k, seed=seed, max_tree_waves=waves,
                    net_mode=net_mode)
    eng.load_random_weights(0)
    sp = oa.SelfPlay(eng)
    sp.reset()
    return eng, sp


@pytest.mark.parametrize("n,count,k", [(9, 100, 8), (15, 100, 16)])
def test_one_wave_is_the_sequential_executor(n, count, k):
    # (row-independent net mode on the sequential side: the shared-tree rounds evaluate every request on its own, and at
    #  N = 15 the default mode's sibling differences carry ~5e-5 of batch-dependent rounding, enough to flip a PUCT comparison)
    a_eng, a = _engine(n, k, 5, 0, net_mode=oa.binding.NET_F16X3_ROWS)
    b_eng, b = _engine(n, k, 5, 1)
    for ply in range(10):
        a.execute(count, k)
        b.execute_shared(count, k, waves=1)
        for side in (0, 1):
            ai, af = a.tree_dump(0, side)
            bi, bf = b.tree_dump(0, side)
            assert np.array_equal(ai, bi) and np.array_equal(af.view(np.uint32), bf.view(np.uint32)), f"ply {ply} side {side}"
            assert a.tree_root(0, side) == b.tree_root(0, side)
        assert np.array_equal(a.sample_actions(1.0, 4), b.sample_actions(1.0, 4))
        a.advance()
        b.advance()
        if a.alive_count == 0:
            break
    assert a_eng.stats()["sims"] == b_eng.stats()["sims"]
    a_eng.close()
    b_eng.close()


@pytest.mark.parametrize("n,count,k,waves", [(9, 400, 8, 8), (9, 256, 4, 16), (15, 800, 16, 16)])
def test_many_waves_keep_the_tree_consistent(n, count, k, waves):
    eng, sp = _engine(n, k, 11, waves, max_nodes=8192)
    hw = n * n
    plies = 0
    while sp.alive_count > 0 and plies < 12:
        side = sp.ply & 1
        before = eng.stats()
        n0 = sp.tree_root(0, side)[2]
        sp.execute_shared(count, k, waves=waves)
        after = eng.stats()
        ints, floats = sp.tree_dump(0, side)
        root_n, root_w, n_nodes, n_tables = sp.tree_root(0, side)
        check_tree_invariants(ints, floats, root_n, hw)
        assert len(ints) == n_nodes
        parent, nch, nvis, status = ints[:, 0], ints[:, 5], ints[:, 6], ints[:, 2]
        sims = after["sims"] - before["sims"]
        evals = after["evals"] - before["evals"]
        assert sims == -(-count // k) * k
        csum = np.zeros(n_nodes, dtype=np.int64)
        np.add.at(csum, parent[1:], nvis[1:])
        visits = np.concatenate([[root_n], nvis[1:]]).astype(np.int64)
        assert np.all(visits >= csum), "a node was visited less often than its children together"
        new_nodes = n_nodes - n0
        assert new_nodes <= sims        # at most one expansion per simulation (duplicate picks are dropped)
        assert evals <= new_nodes       # requests are new nodes; terminal children are not evaluated
        pending = np.flatnonzero(((ints[:, 7] >> 16) == 0) & (status == 0))
        assert len(pending) == 0, "a non-terminal node was left without its evaluation"
        a, cn, cw, cp = sp.root_children(0, side)
        assert len(set(a.tolist())) == len(a) == nch[0]   # no slot of the root's table was claimed twice
        acts = sp.sample_actions(1.0, 30)
        assert 0 <= acts[0] < hw
        sp.advance()
        plies += 1
    assert plies >= 6
    eng.close()


def test_many_waves_search_like_the_sequential_search():
    """Statistical parity (the reference's executor is nondeterministic by design, so there is nothing exact to compare):
    8 waves x K = 8 put 64 simulations between two scatters, like the sequential executor with batch_size 64 and the same
    RNG indices; concurrent waves additionally drop simulations that picked the same untried action (mcts_executor.rs:171-178).
    Over 6 seeds the 8-wave search must put most of its visits where the sequential searches put theirs: the visit mass on
    the sequential search's ten most visited moves, and the total-variation distances (printed; measured 0.17 against the
    K = 64 sequential search, 0.13 between the K = 8 and K = 64 sequential searches themselves)."""
    n, count = 9, 384

    def visits(seed, k, waves):
        eng, sp = _engine(n, k, seed, waves)
        if waves:
            sp.execute_shared(count, k, waves=waves)
        else:
            sp.execute(count, k)
        pi, has = sp.compute_policy()
        eng.close()
        assert has[0]
        return pi[0].astype(np.float64)

    tv_same, tv_other, mass_par, mass_seq = [], [], [], []
    for seed in range(6):
        par = visits(seed, 8, 8)
        seq64 = visits(seed, 64, 0)
        seq8 = visits(seed, 8, 0)
        tv_same.append(0.5 * np.abs(par - seq64).sum())
        tv_other.append(0.5 * np.abs(seq8 - seq64).sum())
        top = np.argsort(-seq8)[:10]
        mass_par.append(par[top].sum())
        mass_seq.append(seq64[top].sum())
    print(f"TV(8 waves x K=8 vs sequential K=64) = {np.mean(tv_same):.3f}   TV(sequential K=8 vs K=64) = {np.mean(tv_other):.3f}   "
          f"mass on the sequential K=8 top-10: 8 waves {np.mean(mass_par):.3f}, sequential K=64 {np.mean(mass_seq):.3f}")
    assert np.mean(tv_same) < 0.35
    assert np.mean(mass_par) > 0.6 * np.mean(mass_seq)


def _dump(sp, side):
    ints, floats = sp.tree_dump(0, side)
    ints = ints.copy()
    ints[:, 7] &= 0xFFFF  # (the literal oracle's dump has no has_policy bit: its placeholder rows are explicit)
    return ints, floats


@pytest.mark.parametrize("n,count,k,waves,plies", [(9, 96, 8, 1, 6), (9, 256, 8, 8, 8), (9, 250, 4, 16, 6), (15, 800, 16, 16, 3)])
def test_recorded_interleaving_replays_exactly_on_the_literal_oracle(n, count, k, waves, plies):
    """The engine searches one tree with `waves` wavefronts in recorded mode; the literal oracle (pointer nodes, stored p, explicit
    refresh loops) replays the recorded lock order with the engine's net outputs: canonical tree dumps, root statistics, request
    boards, sampled moves and the mirror step must be identical.  count = 250 with K = 4 also covers a last group with fewer rounds
    than waves and a count that is not a multiple of K."""
    seed = 13
    eng, sp = _engine(n, k, seed, waves, max_nodes=8192)
    lit = O.Literal(n, 1, seed=seed, cap_nodes=8192)
    lit.reset(eng.evaluate_p(O.Environment(n).encode_nn_input(0)[None]).reshape(-1))
    interleaved = 0
    for ply in range(plies):
        if sp.alive_count == 0:
            break
        side = sp.ply & 1
        groups = sp.execute_shared_recorded(count, k, waves=waves)
        rounds = -(-count // k)
        assert len(groups) == -(-rounds // waves)
        assert sum(len(g[0]) for g in groups) == rounds * k  # every simulation of every round took its turn
        for so, bo, p, v in groups:
            assert len(bo) == len(v) == len(p)
            if np.any(np.diff(so.astype(np.int32)) < 0):
                interleaved += 1  # (a wave with a larger index ran before one with a smaller index: the rounds really interleave)
        lit.shared_run(count, k, 0.25, 0.03, waves, groups)
        gi, gf = _dump(sp, side)
        li, lf = lit.tree_dump(0, side)
        assert gi.shape == li.shape, f"ply {ply}: {gi.shape[0]} vs {li.shape[0]} nodes"
        assert np.array_equal(gi, li), f"ply {ply}: node records differ"
        assert np.array_equal(gf.view(np.uint32), lf.view(np.uint32)), f"ply {ply}: w / policy bits differ"
        a = sp.sample_actions(1.0, 4)
        assert np.array_equal(a, lit.sample(1.0, 4)), f"ply {ply}: moves"
        nm = sp.mirror_generate()
        lm, _ = lit.mirror_generate()
        assert nm == len(lm) and np.array_equal(sp.mirror_inputs(), lm)
        pm = sp.mirror_eval()
        sp.mirror_apply()
        lit.advance(pm)
        for s2 in (0, 1):
            gi, gf = _dump(sp, s2)
            li, lf = lit.tree_dump(0, s2)
            assert np.array_equal(gi, li) and np.array_equal(gf.view(np.uint32), lf.view(np.uint32)), f"ply {ply} after advance, side {s2}"
    if waves > 1:
        assert interleaved > 0, "the recorded schedules were all sequential: nothing concurrent was tested"
    assert lit.error == 0
    eng.close()


def test_recorded_single_wave_equals_the_free_running_one():
    """waves = 1: recorded mode is the same sequential schedule as omok_execute_shared(waves = 1) -- which the first test of this file
    ties to omok_execute -- so the oracle-replayed arithmetic is the arithmetic of the unrecorded kernels too."""
    n, count, k = 15, 112, 16
    a_eng, a = _engine(n, k, 5, 1)
    b_eng, b = _engine(n, k, 5, 1)
    for ply in range(4):
        a.execute_shared(count, k, waves=1)
        b.execute_shared_recorded(count, k, waves=1)
        for side in (0, 1):
            ai, af = a.tree_dump(0, side)
            bi, bf = b.tree_dump(0, side)
            assert np.array_equal(ai, bi) and np.array_equal(af.view(np.uint32), bf.view(np.uint32)), f"ply {ply} side {side}"
        assert np.array_equal(a.sample_actions(1.0, 4), b.sample_actions(1.0, 4))
        a.advance()
        b.advance()
    a_eng.close()
    b_eng.close()
