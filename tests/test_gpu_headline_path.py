"""GPU tests of the path the headline number is measured on ("weak" #2): `omok_selfplay_run` / `omok_execute` fuse
the net's softmax / tanh into the policy scatter (k_softmax_scatter_policy), defer a round's backups into the head of the next
round's kernel, and at board_size 15 evaluate sibling requests as one base position + 7x7-window differences with a per-game cache of
base evaluations.  The oracle tests drive the step-wise API (separate softmax, immediate backups); this file closes the chain:

  (a) omok_execute == the step-wise path, tree dumps bit for bit, at 15x15 / 800 simulations (fully expanded nodes, depth >= 3)
  (b) omok_selfplay_run twice at 15x15 on the difference path -> identical packed replay bytes (slots are handed out by atomics)
  (c) the difference path's p / v of real rounds against the ORACLE's forward (not the fp32 kernels), random-init and trained weights
  (d) base cache on / off -> bit-identical p / v and trees
  (e) rounds of different sizes inside one ply (difference -> copy -> difference path) never read a stale base
Reference: alpha-zero/src/parallel_mcts_executor.rs:194-265 (evaluate + scatter of a round), agent_model.rs:116-134."""
import numpy as np
import pytest

import omok_ai_amd as oa
from omok_ai_amd import binding as B
from oracle import oracle as O
from helpers import trained_tensors, tree_shape

pytestmark = pytest.mark.gpu
TOL = 1e-3


def _dumps(sp, games):
    return [sp.tree_dump(g, s) for g in range(games) for s in (0, 1)]


def _same_dumps(a, b, tag):
    for i, ((ai, af), (bi, bf)) in enumerate(zip(a, b)):
        assert ai.shape == bi.shape and np.array_equal(ai, bi), f"{tag}: node records of tree {i} differ"
        assert np.array_equal(af.view(np.uint32), bf.view(np.uint32)), f"{tag}: w / policy bits of tree {i} differ"


@pytest.mark.parametrize("mode", [B.NET_F16X3_ROWS, B.NET_F32, B.NET_F16X3])
def test_execute_equals_stepwise_in_the_benchmark_regime(mode):
    """(a) N = 15, 2 games, 800 simulations per move, 2 plies: the run loop's rounds (fused softmax + scatter, backups deferred into
    the next k_round) leave the trees the step-wise rounds leave, and those trees are in the regime the headline runs in."""
    n, games, count, k = 15, 2, 800, 16
    tensors = oa.weights.init_random(n, seed=0)
    out = []
    for variant in ("execute", "stepwise"):
        eng = oa.Engine(board_size=n, games=games, max_nodes=4224, max_tables=1056, max_batch_k=k, seed=5, net_mode=mode)
        eng.load_weights(tensors)
        sp = oa.SelfPlay(eng)
        sp.reset()
        per_ply = []
        for _ in range(2):
            if variant == "execute":
                sp.execute(count, k)
                per_ply.append(_dumps(sp, games))
                sp.sample_actions(1.0, 30)
                sp.advance()
            else:
                for rnd in range(count // k):
                    sp.round_generate(rnd, k)
                    sp.round_eval()
                    sp.round_scatter()
                per_ply.append(_dumps(sp, games))
                sp.sample_actions(1.0, 30)
                sp.mirror_generate()
                sp.mirror_eval()
                sp.mirror_apply()
        per_ply.append(_dumps(sp, games))
        out.append(per_ply)
        eng.close()
    for ply, (a, b) in enumerate(zip(out[0], out[1])):
        _same_dumps(a, b, f"mode {mode} after ply {ply}")
    shapes = [tree_shape(ints) for ints, _ in out[0][0]]
    full_nr, depth = max(s[1] for s in shapes), max(s[2] for s in shapes)
    print(f"mode {mode}: fully expanded non-root nodes {full_nr}, depth {depth}")
    assert full_nr >= 1 and depth >= 3


def _packed_run(n, games, count, k, plies, seed, tensors, cache=True, rects=True):
    import torch
    eng = oa.Engine(board_size=n, games=games, max_nodes=4 * count + 256, max_tables=count + 64, max_batch_k=k, seed=seed)
    eng.load_weights(tensors)
    eng.set_base_cache(cache)
    eng.set_window_rects(rects)
    sp = oa.SelfPlay(eng)
    sp.reset()
    sp.run(count, k, max_plies=plies)
    rec = sp.replay_record_bytes()
    cap = games * plies
    buf = torch.zeros(cap * rec, dtype=torch.uint8, device="cuda")
    got = sp.replay_pack_into(buf.data_ptr(), cap)
    data = buf[: got * rec].cpu().numpy().copy()
    dumps = _dumps(sp, min(games, 8))
    eng.close()
    return got, data, dumps


@pytest.mark.parametrize("n,games,count,k,plies", [(15, 256, 96, 16, 4), (9, 256, 48, 8, 5)])
def test_selfplay_run_on_the_difference_path_is_reproducible_and_cache_independent(n, games, count, k, plies):
    """(b) + (d): two runs of omok_selfplay_run with rounds on the difference path (N = 15: 4096 rows >= 3072; N = 9: 2048 rows >= 1024;
    k_group hands out slots with atomics) give the same packed replay bytes and trees; a third run with the base cache switched off
    gives them too (a cached base evaluation == its recomputation, bit for bit); so does a fourth whose fc0 window tiles walk the whole 7x7 window
    instead of the rectangle their rows can differ in (round 5: the skipped window pixels hold exact zeros)."""
    tensors = oa.weights.init_random(n, seed=0)
    a = _packed_run(n, games, count, k, plies, 3, tensors)
    b = _packed_run(n, games, count, k, plies, 3, tensors)
    c = _packed_run(n, games, count, k, plies, 3, tensors, cache=False)
    assert a[0] == games * plies
    assert a[0] == b[0] and np.array_equal(a[1], b[1]), "two identical runs differ"
    _same_dumps(a[2], b[2], "run vs run")
    assert a[0] == c[0] and np.array_equal(a[1], c[1]), "base cache on / off differ"
    _same_dumps(a[2], c[2], "cache on vs off")
    d = _packed_run(n, games, count, k, plies, 3, tensors, rects=False)
    assert a[0] == d[0] and np.array_equal(a[1], d[1]), "window rectangles on / off differ"
    _same_dumps(a[2], d[2], "window rectangles on vs off")


@pytest.mark.parametrize("n,games,k", [(15, 224, 16), (9, 160, 8)])  # 3584 rows per round >= 3072; 1280 >= 1024
@pytest.mark.parametrize("weights", ["random-init", "trained"])
def test_difference_path_outputs_against_the_oracle(weights, n, games, k):
    """(c) the p / v that rounds on the difference path deliver, against the oracle's fp32 forward of the same request rows (>= 256
    rows per weight set, taken across the rounds of two plies), in the default mode with its committed operand format."""
    tensors = oa.weights.init_random(n, seed=2) if weights == "random-init" else trained_tensors(n, 1)[0]
    eng = oa.Engine(board_size=n, games=games, max_nodes=512, max_tables=128, max_batch_k=k, seed=21)
    eng.load_weights(tensors)
    fmt = B.FC0_FORMATS[int(eng.stats()["fc0_format"])]
    net = O.Net(n, tensors)
    sp = oa.SelfPlay(eng)
    sp.reset()
    rng = np.random.default_rng(0)
    rows, dp, dv, differing = 0, 0.0, 0.0, 0
    for ply in range(2):
        for rnd in range(5):
            nreq = sp.round_generate(rnd, k, 0.25, 0.03)
            x = sp.round_inputs().copy()
            p, v = sp.round_eval()
            p, v = np.array(p).reshape(nreq, -1).copy(), np.array(v).reshape(-1).copy()
            sp.round_scatter()
            if rnd == 0:
                continue
            assert nreq == games * k
            pick = rng.choice(nreq, size=48, replace=False)
            pc, vc = net.forward(x[pick], threads=8)
            dp, dv = max(dp, float(np.abs(p[pick] - pc).max())), max(dv, float(np.abs(v[pick] - vc).max()))
            pp, _ = eng.evaluate_pv(x[pick])
            differing += int((p[pick].view(np.uint32) != pp.reshape(len(pick), -1).view(np.uint32)).any(axis=1).sum())
            rows += len(pick)
        sp.sample_actions(1.0, 30)
        sp.advance()
    print(f"difference path vs the oracle, {weights} weights (format {fmt}): {rows} rows, max|dp| {dp:.2e} max|dv| {dv:.2e}; {differing} rows differ from row-by-row bits")
    assert rows >= 256 and dp < TOL and dv < TOL
    assert differing > rows // 2  # (the path under test really ran)
    eng.close()


@pytest.mark.parametrize("mode", [B.NET_F16X3_FP6, B.NET_F16X3_F16])
def test_rounds_of_different_sizes_in_one_ply_never_read_a_stale_base(mode):
    """(e) K = 16, 4, 16 rounds inside one ply at 400 games: 6400 rows (difference path: bases cached per game), 1600 rows (copy
    path: it keeps its h grids in the slots the cache uses), 6400 rows again.  Every round's p / v must be those of the requested
    positions: the copy path equals row-by-row evaluation bit for bit, the difference path stays within 5e-4 of it.  Also: the copy
    path's operand rows == the rows a row-by-row evaluation writes, byte for byte (both operand formats)."""
    n, games = 15, 400
    eng = oa.Engine(board_size=n, games=games, max_nodes=512, max_tables=128, max_batch_k=16, seed=8, net_mode=mode)
    eng.load_random_weights(3)
    sp = oa.SelfPlay(eng)
    sp.reset()
    for ply in range(2):
        for rnd, k in enumerate((16, 16, 4, 16, 4, 16)):
            nreq = sp.round_generate(rnd, k, 0.25, 0.03)
            x = sp.round_inputs().copy()
            p, v = sp.round_eval()
            p, v = np.array(p).reshape(nreq, -1).copy(), np.array(v).reshape(-1).copy()
            copy_path = games * k < 3072
            rows_round = eng.operand_rows(0, min(nreq, 640)).copy() if copy_path else None
            sp.round_scatter()
            pp, vp = eng.evaluate_pv(x)
            pp, vp = pp.reshape(nreq, -1), vp.reshape(-1)
            if copy_path:
                assert np.array_equal(p.view(np.uint32), pp.view(np.uint32)) and np.array_equal(v.view(np.uint32), vp.view(np.uint32)), (ply, rnd)
                live = rows_round.shape[1] - 80 * 16  # (the row pad is never written)
                assert np.array_equal(rows_round[:, :live], eng.operand_rows(0, min(nreq, 640))[:, :live]), (ply, rnd)
            else:
                assert np.abs(p - pp).max() < 5e-4 and np.abs(v - vp).max() < 5e-4, (ply, rnd, float(np.abs(p - pp).max()), float(np.abs(v - vp).max()))
        sp.sample_actions(1.0, 30)
        sp.advance()
    eng.close()


@pytest.mark.parametrize("mode", [B.NET_F16X3_F16, B.NET_F16X3_FP6])
@pytest.mark.parametrize("n,games,k", [(15, 224, 16), (9, 160, 8)])
def test_both_children_kernels_of_the_difference_path_agree(n, games, k, mode):
    """The difference path evaluates a run's children with k_sib_children2 (one wave per child, windows that grow with the blocks, the base's depthwise
    outputs + the depthwise of the difference); omok_debug_set_children_kernel(1) keeps k_sib_children (wave pair per child, the whole 7x7 window
    through every block, halo ring from the base).  Same requests -> p / v within 2e-4 of each other and each within 1e-3 of the oracle; the engine's launch
    counters say which kernel ran.  (The outputs are usually bit-identical: every layer boundary re-quantises to f16 hi + lo, ~22 bits, which absorbs the
    1e-7-level differences of the two depthwise summation orders -- 10752 rows of the first run of this test did not differ in one bit.)"""
    tensors = oa.weights.init_random(n, seed=4)
    net = O.Net(n, tensors)
    outs = []
    for which in (2, 1):
        eng = oa.Engine(board_size=n, games=games, max_nodes=512, max_tables=128, max_batch_k=k, seed=13, net_mode=mode)
        eng.load_weights(tensors)
        eng.set_children_kernel(which)
        sp = oa.SelfPlay(eng)
        sp.reset()
        per = []
        for rnd in range(4):
            nreq = sp.round_generate(rnd, k, 0.25, 0.03)
            x = sp.round_inputs().copy()
            p, v = sp.round_eval()
            per.append((x, np.array(p).reshape(nreq, -1).copy(), np.array(v).reshape(-1).copy()))
            sp.round_scatter()
        outs.append(per)
        st = eng.stats()
        assert (st["children2_launches"], st["children1_launches"]) == ((4.0, 0.0) if which == 2 else (0.0, 4.0)), st
        eng.close()
    rng = np.random.default_rng(1)
    differing = 0
    for rnd in range(1, 4):
        (xa, pa, va), (xb, pb, vb) = outs[0][rnd], outs[1][rnd]
        assert np.array_equal(xa, xb)
        assert np.abs(pa - pb).max() < 2e-4 and np.abs(va - vb).max() < 2e-4, (rnd, float(np.abs(pa - pb).max()), float(np.abs(va - vb).max()))
        differing += int((pa.view(np.uint32) != pb.view(np.uint32)).any(axis=1).sum())
        pick = rng.choice(len(xa), size=32, replace=False)
        pc, vc = net.forward(xa[pick], threads=8)
        for p, v in ((pa, va), (pb, vb)):
            assert np.abs(p[pick] - pc).max() < TOL and np.abs(v[pick] - vc).max() < TOL
    print(f"children kernels 2 vs 1, n={n} mode {mode}: {differing} rows differ in bits")


def test_execute_equals_stepwise_on_the_difference_path_with_deep_trees():
    """(a) x the difference path: N = 15, 256 games (4096-row rounds >= 3072: sibling base + window differences, base cache, slots
    handed out by atomics) x 800 simulations x 2 plies in the DEFAULT net mode: omok_execute (fused softmax + scatter, deferred backups, k_scan zeroing the
    grouping counters, k_group writing the request list) leaves the trees the step-wise rounds leave, bit for bit, and a second run of omok_execute leaves them
    again.  Compared: root visit policies and sampled moves of ALL games after every ply, full canonical dumps of 16 trees (first and last games)."""
    n, games, count, k = 15, 256, 800, 16
    tensors = oa.weights.init_random(n, seed=0)
    picked = list(range(4)) + list(range(games - 4, games))

    def play(variant):
        eng = oa.Engine(board_size=n, games=games, max_nodes=4224, max_tables=1056, max_batch_k=k, seed=17)
        eng.load_weights(tensors)
        sp = oa.SelfPlay(eng)
        sp.reset()
        per_ply = []
        for _ in range(2):
            if variant == "stepwise":
                for rnd in range(count // k):
                    assert sp.round_generate(rnd, k) >= 3072
                    sp.round_eval()
                    sp.round_scatter()
            else:
                sp.execute(count, k)
            pol = sp.compute_policy()[0].copy()
            dumps = [sp.tree_dump(g, s) for g in picked for s in (0, 1)]
            acts = np.array(sp.sample_actions(1.0, 30)).copy()
            if variant == "stepwise":
                sp.mirror_generate()
                sp.mirror_eval()
                sp.mirror_apply()
            else:
                sp.advance()
            per_ply.append((pol, acts, dumps))
        st = eng.stats()
        eng.close()
        return per_ply, st

    a, st_a = play("execute")
    b, _ = play("stepwise")
    c, _ = play("execute")
    assert st_a["children2_launches"] >= 2 * (count // k) - 2, st_a  # (the rounds really took the difference path)
    for ply in range(2):
        for other, tag in ((b, "execute vs step-wise"), (c, "execute vs execute")):
            assert np.array_equal(a[ply][0].view(np.uint32), other[ply][0].view(np.uint32)), f"{tag}: visit policies differ after ply {ply}"
            assert np.array_equal(a[ply][1], other[ply][1]), f"{tag}: sampled moves differ at ply {ply}"
            _same_dumps(a[ply][2], other[ply][2], f"{tag}, ply {ply}")
    shapes = [tree_shape(ints) for ints, _ in a[1][2]]
    print(f"difference path, 800 sims: fully expanded non-root nodes {max(s[1] for s in shapes)}, depth {max(s[2] for s in shapes)}")
    assert max(s[1] for s in shapes) >= 1 and max(s[2] for s in shapes) >= 3


def test_whole_episodes_at_scale_are_deterministic():
    """The determinism soak of tools/soak_determinism.py as a test: two fresh engines play the same whole episode -- 1024 games at 15x15, 512 simulations per
    move (16384-row rounds: multi-tile window bins, the K-split set, the base cache, every thin-round path of the tail) -- and the digests of their packed replay
    records are equal.  A data race in any kernel of the run loop shows up here."""
    import hashlib
    import torch
    n, games, sims, k = 15, 1024, 512, 16
    digests = []
    for run in range(2):
        eng = oa.Engine(board_size=n, games=games, max_nodes=4 * sims + 1024, max_tables=(4 * sims + 1024) // 4, max_batch_k=k, seed=123)
        eng.load_random_weights(0)
        sp = oa.SelfPlay(eng)
        sp.reset()
        st = sp.run(sims, k, 0.25, 0.03, 1.0, 30, 0)
        _, _, plies = sp.game_info()
        rec, total = sp.replay_record_bytes(), int(plies.sum())
        buf = torch.zeros(total * rec, dtype=torch.uint8, device="cuda:0")
        assert sp.replay_pack_into(buf.data_ptr(), total) == total
        digests.append(hashlib.sha256(buf.cpu().numpy().tobytes()).hexdigest())
        print(f"run {run}: games {int(st['finished'])} plies {total} sims {int(st['sims'])} sha256 {digests[-1][:16]}")
        assert st["finished"] == games and st["children2_launches"] > 0
        eng.close()
    assert digests[0] == digests[1], digests


@pytest.mark.parametrize("n,games,count,k,max_tables", [(15, 4096, 800, 16, 1056), (9, 16384, 200, 8, 456)])
def test_full_size_rounds_first_games_against_the_oracle(n, games, count, k, max_tables):
    """Tree parity AT THE TIMED SIZES (BASELINE configs[1]: 4096 games, 15x15, 800 simulations per move, K = 16; configs[2]: 16384 games, 9x9, 200 simulations, K = 8): every
    round is a 65536-row (131072-row) forward on the difference path
    -- multi-tile window bins, cost-ordered window tiles with rectangles, the K-split set -- and the first 32 games are played in step on the ORACLE, which consumes the GPU's
    p / v rows of those games (the dense request list is in tree order: the first rows of every round).  Request boards (sampled rounds), moves, mirror inputs and the canonical
    dumps of both trees of the 32 games must be bit-identical after one whole ply (50 rounds: fully expanded nodes) and after ten rounds of the second.
    Reference: alpha-zero/src/parallel_mcts_executor.rs:26-270, agent.rs:83-232, src/trainer.rs:95-205."""
    from test_gpu_parity import _compare_trees
    g0 = 32
    tensors = oa.weights.init_random(n, seed=0)
    eng = oa.Engine(board_size=n, games=games, max_nodes=4 * count + 1024, max_tables=max_tables, max_batch_k=k, seed=0)
    eng.load_weights(tensors)
    sp = oa.SelfPlay(eng)
    sp.reset()
    root_p = eng.evaluate_p(O.Environment(n).encode_nn_input(0)[None]).reshape(-1)
    osp = O.SelfPlay(n, g0, cap_nodes=4 * count + 1024, cap_tables=max_tables, seed=0, game_offset=0)
    osp.reset(root_p)
    onet = O.Net(n, tensors)
    eng.reset_stats()
    shape = [0, 0]
    for ply, rounds in ((0, count // k), (1, 10)):  # (a whole ply: 50 / 25 rounds; then ten rounds of the second)
        for rnd in range(rounds):
            nreq = sp.round_generate(rnd, k, 0.25, 0.03)
            oin = osp.round_generate(rnd, k, 0.25, 0.03)
            assert nreq == games * k  # (no terminal position this early: every simulation asks for an evaluation)
            xin = None
            if rnd in (0, 1, 17, rounds - 1):
                xin = sp.round_inputs()
                assert np.array_equal(xin[: len(oin)], oin), f"ply {ply} round {rnd}: request boards of the first {g0} games"
            p, v = sp.round_eval()
            if xin is not None and rnd == 17:  # the net's outputs of a full-size round, rows from all over the batch, against the oracle's forward (north_star: 1e-3, logits included)
                pick = np.random.default_rng(5).choice(nreq, size=384, replace=False)
                lg, vp = sp.round_logits()
                pc, vc, lgc, vpc = onet.forward_logits(xin[pick], threads=8)
                worst = (float(np.abs(p[pick] - pc).max()), float(np.abs(v[pick] - vc).max()), float(np.abs(lg[pick] - lgc).max()), float(np.abs(vp[pick] - vpc).max()))
                print(f"n={n}: full-size round {rnd}, {len(pick)} of {nreq} rows against the oracle: |dp| {worst[0]:.2e} |dv| {worst[1]:.2e} |dlogit| {worst[2]:.2e} |dvpre| {worst[3]:.2e}")
                assert max(worst) < TOL, worst
            sp.round_scatter()
            osp.round_scatter(p[: len(oin)], v[: len(oin)])
        _compare_trees(sp, osp, g0, f"ply {ply} after {rounds} full-size rounds")
        for g in range(g0):
            full, full_nr, depth = tree_shape(osp.tree_dump(g, ply & 1)[0])
            shape = [max(shape[0], full_nr), max(shape[1], depth)]
        if ply == 0:
            a = sp.sample_actions(1.0, 30)
            assert np.array_equal(a[:g0], osp.sample(1.0, 30)), "moves of the first games"
            nm = sp.mirror_generate()
            om = osp.mirror_generate()
            assert nm == games and np.array_equal(sp.mirror_inputs()[: len(om)], om)
            pm = sp.mirror_eval()
            sp.mirror_apply()
            osp.advance(pm[: len(om)])
            _compare_trees(sp, osp, g0, "after the first advance")
    st = eng.stats()
    print(f"full-size rounds: {int(st['children2_launches'])} on the difference path, fully expanded non-root nodes {shape[0]}, depth {shape[1]}")
    assert st["children2_launches"] >= count // k + 10 and shape[0] >= 1 and shape[1] >= 2
    eng.close()


def test_executed_work_counters_account_for_every_request_row():
    """OMOK_STAT_WORK_* (the device-side counters behind bench.py's `executed_flops`): over a few plies of omok_selfplay_run with rounds on BOTH sibling paths (difference
    path while >= 2048 rows, copy path below) every request row of a search round is counted exactly once -- as a child of a run or as a single row, on one of the two
    paths -- runs evaluated in full never exceed the runs, and the window tiles walk between 1 and 49 pixels each."""
    n, k, sims = 15, 16, 64
    for games, diff_expected in ((192, True), (48, False)):  # 3072-row rounds (difference path), 768-row rounds (copy path)
        eng = oa.Engine(board_size=n, games=games, max_nodes=1024, max_tables=256, max_batch_k=k, seed=5)
        eng.load_random_weights(0)
        sp = oa.SelfPlay(eng)
        sp.reset()
        eng.reset_stats()
        st = sp.run(sims, k, 0.25, 0.03, 1.0, 30, 3)
        rounds = 3 * (sims // k)
        diff_rows = st["work_diff_children"] + st["work_diff_singles"]
        copy_rows = st["work_copy_children"] + st["work_copy_singles"]
        # every simulation of these first plies ends in one request row (no terminal leaves yet); the engine's `evals` adds the plies' mirror evaluations (at most one per game and ply)
        assert diff_rows + copy_rows == st["sims"] == rounds * k * games, (games, st)
        assert diff_rows + copy_rows <= st["evals"] <= diff_rows + copy_rows + 3 * games, (games, st)
        assert (st["children2_launches"] > 0) == diff_expected and (diff_rows > 0) == diff_expected
        assert st["children2_launches"] + st["children1_launches"] == rounds
        if diff_expected:
            assert 0 < st["work_diff_full_runs"] <= st["work_diff_runs"] and st["work_diff_children"] >= 3 * st["work_diff_runs"]
            assert st["work_win_tiles"] <= st["work_win_pixels"] <= 49 * st["work_win_tiles"]
            assert 1 <= st["work_full_tiles"] <= 4 * st["children2_launches"]  # (a round whose bases are all cached and that has no single rows evaluates nothing in full)
        else:
            assert st["work_copy_children"] >= 3 * st["work_copy_runs"] > 0 and st["work_win_pixels"] == 0
        eng.reset_stats()
        assert eng.stats()["work_diff_children"] == 0 and eng.stats()["work_copy_children"] == 0
        eng.close()
