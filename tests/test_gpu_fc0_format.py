"""GPU tests of fc0's operand format: the 1e-3 contract of BASELINE.json's north_star ("policy/value within 1e-3 of the
reference CPU path") must hold BY DEFAULT -- on random-init, scaled, heavy-tailed and TRAINED weights, for both board sizes -- because
omok_net_commit measures the fast format (block-scaled fp6 correction terms) on a probe set and falls back to f16 correction terms
when its worst |dp|, |dv| exceed 3e-4.  The checker is the oracle's fp32 forward (oracle/net.c); the OMOK_NET_F32 kernels are a second
reference.  Contract quantities: p after softmax, v after tanh (AgentModel::evaluate_pv, alpha-zero/src/agent_model.rs:116-134); the
pre-softmax logits / pre-tanh value are reported beside them."""
import os

import numpy as np
import pytest

import omok_ai_amd as oa
from omok_ai_amd import binding as B
from oracle import oracle as O
from helpers import random_positions, trained_tensors

pytestmark = pytest.mark.gpu
TOL = 1e-3
LIMIT = 3e-4  # NET_PROBE_LIMIT


def _report(n, tensors, x, tag, mode=B.NET_F16X3, via_file=None):
    eng = oa.Engine(board_size=n, games=32, max_nodes=16, max_tables=8, max_batch_k=16, net_mode=mode)
    ref = oa.Engine(board_size=n, games=32, max_nodes=16, max_tables=8, max_batch_k=16, net_mode=B.NET_F32)
    if via_file:
        eng.load(via_file)
    else:
        eng.load_weights(tensors)
    ref.load_weights(tensors)
    st = eng.stats()
    p, v = eng.evaluate_pv(x)
    lg, vp = eng.evaluate_logits(x)
    p32, v32 = ref.evaluate_pv(x)
    lg32, vp32 = ref.evaluate_logits(x)
    pc, vc = O.Net(n, tensors).forward(x, threads=8)
    p, p32 = p.reshape(len(x), -1), p32.reshape(len(x), -1)
    out = {"dp_oracle": float(np.abs(p - pc).max()), "dv_oracle": float(np.abs(v.reshape(-1) - vc).max()),
           "dp_f32": float(np.abs(p - p32).max()), "dv_f32": float(np.abs(v - v32).max()),
           "dlogit": float(np.abs(lg - lg32).max()), "dvpre": float(np.abs(vp - vp32).max()), "logit_max": float(np.abs(lg32).max()),
           "f32_vs_oracle_dp": float(np.abs(p32 - pc).max()), "format": B.FC0_FORMATS[int(st["fc0_format"])],
           "probe": (st["probe_rows"], st["probe_dp_fp6"], st["probe_dv_fp6"], st["probe_dp_f16"], st["probe_dv_f16"]), "stats": st}
    print(f"precision[{tag}] " + " ".join(f"{k}={val:.3e}" if isinstance(val, float) else f"{k}={val}" for k, val in out.items()))
    eng.close()
    ref.close()
    return out


def _assert_contract(r, tag):
    assert r["dp_oracle"] < TOL and r["dv_oracle"] < TOL and r["dp_f32"] < TOL and r["dv_f32"] < TOL, (tag, r)


def expected_format(st):
    """omok_net_commit's rule (net_probe) restated on the statistics it publishes: the fastest of fp6 < mixed < f16 whose every probe figure -- plain rows
    and, where the engine's rounds can take the difference path, the synthetic sibling round -- is inside the limits (|dp|, |dv| <= 3e-4, |dlogit| <= 5e-4)."""
    lim, llim = st["probe_limit"], st["probe_logit_limit"]

    def inside(dp, dv, dl):
        return dp <= lim and dv <= lim and dl <= llim
    rounds = st["probe_round_rows"] > 0
    fp6 = inside(st["probe_dp_fp6"], st["probe_dv_fp6"], st["probe_dlogit_fp6"]) and \
        (not rounds or inside(st["probe_round_dp_fp6"], st["probe_round_dv_fp6"], st["probe_round_dlogit_fp6"]))
    mixed = rounds and inside(st["probe_round_dp_mixed"], st["probe_round_dv_mixed"], st["probe_round_dlogit_mixed"])  # (its full rows are the f16 format's)
    return "fp6" if fp6 else "mixed" if mixed else "f16"


def _assert_probe_rule(r):
    rows, dp6, dv6, dp16, dv16 = r["probe"]
    st = r["stats"]
    assert rows >= 512 and abs(st["probe_limit"] - LIMIT) < 1e-9 and abs(st["probe_logit_limit"] - 5e-4) < 1e-9
    assert r["format"] == expected_format(st), r
    if r["format"] == "fp6":  # what was kept has the margin on the probe set, logits included
        assert dp6 <= LIMIT and dv6 <= LIMIT and st["probe_dlogit_fp6"] <= 5e-4
    assert dp16 < LIMIT and dv16 < LIMIT, r  # the fallback itself is well inside


@pytest.mark.parametrize("n", [9, 15])
@pytest.mark.parametrize("mode,name", [(B.NET_F16X3_FP6, "fp6"), (B.NET_F16X3_F16, "f16"), (B.NET_F16X3_MIXED, "mixed")])
def test_forced_formats_on_random_init(n, mode, name):
    """Both formats against the oracle on the random initialiser; the f16 format with a 4x margin."""
    x = random_positions(n, 192, 9)
    r = _report(n, oa.weights.init_random(n, seed=1), x, f"n={n} forced {name}", mode)
    assert r["format"] == name and r["probe"][0] == 0  # (no probe when the format is forced)
    _assert_contract(r, name)
    if name in ("f16", "mixed"):  # (plain rows of the mixed format ARE f16-format rows)
        assert max(r["dp_oracle"], r["dv_oracle"], r["dp_f32"], r["dv_f32"]) < 2.5e-4, r


@pytest.mark.parametrize("n", [9, 15])
def test_commit_probe_chooses_by_measurement(n):
    """omok_get_stats exposes the probe; the format in use follows the documented rule; the default mode then holds the contract on
    other positions than the probe's (random-init seeds 0..2: at N = 9 seed 0 is a net the fp6 format does not pass)."""
    x = random_positions(n, 192, 31)
    formats = []
    for seed in (0, 1, 2):
        r = _report(n, oa.weights.init_random(n, seed=seed), x, f"n={n} seed {seed} auto")
        _assert_probe_rule(r)
        _assert_contract(r, f"seed {seed}")
        formats.append(r["format"])
    print(f"n={n}: formats chosen for seeds 0..2: {formats}")


@pytest.mark.parametrize("n", [9, 15])
def test_contract_on_scaled_and_heavy_tailed_weights(n):
    """Away from the random initialiser: fc0 / policy-head weights x0.5 and x2 (logits scale with the square), a heavy-tailed fc0
    (0.05 % of its entries x50: the worst case for a block-scaled low-precision correction term)."""
    x = random_positions(n, 192, 9)
    base = oa.weights.init_random(n, seed=1)
    variants = {}
    for name, f in (("x0.5", 0.5), ("x2", 2.0)):
        t = [a.copy() for a in base]
        t[23] = t[23] * f
        t[29] = t[29] * f
        variants[name] = t
    heavy = [a.copy() for a in base]
    rng = np.random.default_rng(0)
    idx = rng.choice(heavy[23].size, size=heavy[23].size // 2000, replace=False)
    heavy[23].reshape(-1)[idx] *= 50.0
    variants["heavy-tailed fc0"] = heavy
    for name, t in variants.items():
        r = _report(n, t, x, f"n={n} {name}")
        _assert_probe_rule(r)
        _assert_contract(r, name)


@pytest.mark.parametrize("n", [9, 15])
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_contract_after_training_steps(n, seed):
    """Weights after 200 Adadelta steps on self-play records of the same net, three training seeds per board size: the default
    mode holds 1e-3 on p AND v against the oracle (round 2 measured 1.05e-3 on v at N = 9 with fp6 correction terms and had relaxed
    this test to 2e-3: the bound is back, the kernel choice is what changed)."""
    trained, initial = trained_tensors(n, seed)
    moved = max(float(np.abs(a - b).max()) for a, b in zip(trained, initial))
    assert moved > 1e-3, "training did not move the weights"
    x = random_positions(n, 256, 21 + seed)
    r = _report(n, trained, x, f"n={n} seed {seed} after 200 training steps")
    _assert_probe_rule(r)
    _assert_contract(r, "trained")
    chk = oa.precision.measure(trained, n, x)
    assert abs(chk["max_dv"] - r["dv_f32"]) < 1e-6 and chk["within_contract"]
    if n == 15:  # the search rounds' own path (base + window differences) with the trained weights, on the rows of real rounds
        rounds = oa.precision.measure_search_rounds(trained, n, games=256, batch_k=16, rounds=4, plies=2, seed=9 + seed)
        print(f"n=15 seed {seed} trained, search rounds: {rounds}")
        assert rounds["rows"] > 20000 and rounds["max_dp"] < TOL and rounds["max_dv"] < TOL, rounds
        assert rounds["logits_within_1e-3"], rounds  # (round 4: the logits of the difference path too, in the format the probe chose)


@pytest.mark.parametrize("n", [9, 15])
def test_weights_file_in_the_reference_format_with_larger_magnitudes(n, tmp_path):
    """A weights file in ModelIO's format (alpha-zero/src/model_io.rs:20-24,59-120) whose tensors are larger than any initialiser
    makes them (fc matrices x1.3, convolutions up to x1.1, non-zero biases): omok_net_load_file commits through the same probe."""
    rng = np.random.default_rng(5)
    t = oa.weights.init_random(n, seed=3)
    for i, a in enumerate(t):
        if a.ndim > 1:  # the fc matrices x1.3 (logits x2.2), the convolutions x1.0 .. x1.1; small non-zero biases everywhere
            f = 1.3 if i >= 23 else 1.0 + 0.1 * rng.random()
            t[i] = (a * np.float32(f)).astype(np.float32)
        else:
            t[i] = (0.02 * rng.standard_normal(a.shape)).astype(np.float32)
    path = os.path.join(tmp_path, "alpha-zero")
    oa.model_file.save(path, oa.weights.tensor_names(), t)
    x = random_positions(n, 192, 77)
    r = _report(n, t, x, f"n={n} weights file, larger magnitudes", via_file=path)
    _assert_probe_rule(r)
    _assert_contract(r, "file")


def test_format_switch_between_commits_keeps_results_consistent():
    """One engine, three commits (a net that keeps fp6, one that needs f16, the first again): the row strides, the sibling cache and
    the kernels follow the format of the LAST commit; the results of the first and third commit are bit-identical."""
    n = 9
    x = random_positions(n, 64, 3)
    eng = oa.Engine(board_size=n, games=32, max_nodes=16, max_tables=8, max_batch_k=16)
    outs, fmts = [], []
    for seed in (1, 0, 1):
        eng.load_weights(oa.weights.init_random(n, seed=seed))
        fmts.append(B.FC0_FORMATS[int(eng.stats()["fc0_format"])])
        outs.append(eng.evaluate_pv(x))
    print("formats:", fmts)
    assert np.array_equal(outs[0][0].view(np.uint32), outs[2][0].view(np.uint32)) and np.array_equal(outs[0][1].view(np.uint32), outs[2][1].view(np.uint32))
    assert fmts[0] == fmts[2]
    eng.close()


@pytest.mark.parametrize("n,games,k", [(15, 256, 16), (9, 160, 8)])
@pytest.mark.parametrize("seed", [0, 2])
def test_probe_covers_the_difference_path_and_logits_hold_there(n, games, k, seed):
    """Round 4 ("probe what is actually run"): an engine whose rounds are large enough for the difference path (4096 / 1280 rows) also probes one
    synthetic sibling round -- base rows + window difference rows -- in the fp6, mixed (f16 rows + fp6 differences) and f16 formats and keeps the fastest format whose
    every figure, LOGITS included, is inside the limits.  Then, on real rounds of that engine: p, v AND the pre-softmax logits / pre-tanh value against the oracle's
    forward -- north_star's tolerance as written ("policy/value logits within 1e-3") in the mode the engine chose by itself."""
    tensors = oa.weights.init_random(n, seed=seed)
    eng = oa.Engine(board_size=n, games=games, max_nodes=512, max_tables=128, max_batch_k=k, seed=31)
    eng.load_weights(tensors)
    st = eng.stats()
    fmt = B.FC0_FORMATS[int(st["fc0_format"])]
    assert st["probe_round_rows"] >= 256, st
    assert fmt == expected_format(st), st
    for tag in ("mixed", "f16"):  # the formats with f16 rows are far inside on the synthetic round
        assert st[f"probe_round_dp_{tag}"] < LIMIT and st[f"probe_round_dv_{tag}"] < LIMIT, st
    net = O.Net(n, tensors)
    sp = oa.SelfPlay(eng)
    sp.reset()
    rng = np.random.default_rng(seed)
    rows, dp, dv, dl, dvp, lmax = 0, 0.0, 0.0, 0.0, 0.0, 0.0
    for ply in range(2):
        for rnd in range(4):
            nreq = sp.round_generate(rnd, k, 0.25, 0.03)
            x = sp.round_inputs().copy()
            p, v = sp.round_eval()
            lg, vp = sp.round_logits()
            sp.round_scatter()
            if rnd == 0:
                continue
            pick = rng.choice(nreq, size=48, replace=False)
            pc, vc, lgc, vpc = net.forward_logits(x[pick], threads=8)
            p, v = np.array(p).reshape(nreq, -1), np.array(v).reshape(-1)
            dp, dv = max(dp, float(np.abs(p[pick] - pc).max())), max(dv, float(np.abs(v[pick] - vc).max()))
            dl, dvp = max(dl, float(np.abs(lg[pick] - lgc).max())), max(dvp, float(np.abs(vp[pick] - vpc).max()))
            lmax = max(lmax, float(np.abs(lgc).max()))
            rows += len(pick)
        sp.sample_actions(1.0, 30)
        sp.advance()
    print(f"precision[n={n} seed {seed} difference path, format {fmt}] rows={rows} dp={dp:.2e} dv={dv:.2e} dlogit={dl:.2e} dvpre={dvp:.2e} |logit|max={lmax:.1f} "
          f"probe round: fp6 {st['probe_round_dp_fp6']:.1e}/{st['probe_round_dv_fp6']:.1e}/{st['probe_round_dlogit_fp6']:.1e} "
          f"mixed {st['probe_round_dp_mixed']:.1e}/{st['probe_round_dv_mixed']:.1e}/{st['probe_round_dlogit_mixed']:.1e} "
          f"f16 {st['probe_round_dp_f16']:.1e}/{st['probe_round_dv_f16']:.1e}/{st['probe_round_dlogit_f16']:.1e}")
    assert rows >= 256 and dp < TOL and dv < TOL
    assert dl < TOL and dvp < TOL, "north_star's tolerance on the logits themselves"
    assert eng.stats()["children2_launches"] >= 6  # (the rounds took the difference path)
    eng.close()


def expected_verdict(st):
    """OMOK_STAT_PROBE_OUTSIDE restated (net_probe): the figures of the COMMITTED format -- its plain rows (f16's for mixed / f16) and, where measured, its sibling round --
    0 inside the margin limits, 1 inside north_star's 1e-3 only, 2 outside 1e-3 (the engine then runs the fp32 kernels and reports format f32)."""
    fmt = expected_format(st)
    plain = "fp6" if fmt == "fp6" else "f16"
    figs = [(st[f"probe_dp_{plain}"], st[f"probe_dv_{plain}"], st[f"probe_dlogit_{plain}"])]
    if st["probe_round_rows"] > 0:
        figs.append((st[f"probe_round_dp_{fmt}"], st[f"probe_round_dv_{fmt}"], st[f"probe_round_dlogit_{fmt}"]))
    if all(dp <= st["probe_limit"] and dv <= st["probe_limit"] and dl <= st["probe_logit_limit"] for dp, dv, dl in figs):
        return 0
    return 1 if all(max(f) <= 1e-3 for f in figs) else 2


@pytest.mark.parametrize("n", [9, 15])
def test_probe_verdict_is_published_and_a_net_outside_the_contract_falls_back_to_fp32(n):
    """VERDICT round 4, weak 2 / missing 4: the chooser must never commit a net outside its limits silently, and the ladder needs a safe last rung.
    (a) random-init nets: the verdict follows the published figures (0 or 1; the headline net's f16 plain-row |dlogit| 5.2e-4 is a 1).
    (b) fc matrices x4 (logits x ~64: |logit| in the thousands): even f16 correction terms miss 1e-3 on the logits -> verdict 2, the engine evaluates with the fp32
        kernels (format f32: the arithmetic of AgentModel::evaluate_pv, agent_model.rs:116-134) and its outputs ARE the fp32 engine's, bit for bit.
    (c) a commit of an ordinary net on the same engine returns to the split-precision path with the results a fresh engine gives."""
    x = random_positions(n, 96, 13)
    base = oa.weights.init_random(n, seed=1)
    eng = oa.Engine(board_size=n, games=32, max_nodes=256, max_tables=64, max_batch_k=16)
    eng.load_weights(base)
    st = eng.stats()
    assert int(st["probe_outside"]) == expected_verdict(st) and int(st["probe_outside"]) in (0, 1), st
    fmt0 = B.FC0_FORMATS[int(st["fc0_format"])]
    p0, v0 = eng.evaluate_pv(x)
    big = [a.copy() for a in base]
    for i in (23, 25, 29):
        big[i] = big[i] * np.float32(4.0)
    eng.load_weights(big)
    st = eng.stats()
    print(f"n={n} x4 net: verdict {st['probe_outside']}, format {B.FC0_FORMATS[int(st['fc0_format'])]}, probe f16 |dp| {st['probe_dp_f16']:.2e} |dv| {st['probe_dv_f16']:.2e} "
          f"|dlogit| {st['probe_dlogit_f16']:.2e}, max |logit| {st['probe_logit_max']:.0f}")
    assert int(st["probe_outside"]) == 2 and B.FC0_FORMATS[int(st["fc0_format"])] == "f32", st
    ref = oa.Engine(board_size=n, games=32, max_nodes=256, max_tables=64, max_batch_k=16, net_mode=B.NET_F32)
    ref.load_weights(big)
    p, v = eng.evaluate_pv(x)
    p32, v32 = ref.evaluate_pv(x)
    assert np.array_equal(p.view(np.uint32), p32.view(np.uint32)) and np.array_equal(v.view(np.uint32), v32.view(np.uint32)), "the fallback must be the fp32 kernels"
    lg, vp = eng.evaluate_logits(x)
    lg32, vp32 = ref.evaluate_logits(x)
    assert np.array_equal(lg.view(np.uint32), lg32.view(np.uint32)) and np.array_equal(vp.view(np.uint32), vp32.view(np.uint32))
    pc, vc = O.Net(n, big).forward(x, threads=8)
    assert np.abs(p.reshape(len(x), -1) - pc).max() < TOL and np.abs(v.reshape(-1) - vc).max() < TOL  # ... and it holds the contract against the oracle
    # the fallback also carries a search: two plies of a small self-play, trees equal to those of the fp32 engine
    sp, spr = oa.SelfPlay(eng), oa.SelfPlay(ref)
    for s in (sp, spr):
        s.reset()
        for _ in range(2):
            s.execute(32, 8, 0.25, 0.03)
            s.sample_actions(1.0, 30)
            s.advance()
    for g in range(4):
        for side in (0, 1):
            (ai, af), (bi, bf) = sp.tree_dump(g, side), spr.tree_dump(g, side)
            assert np.array_equal(ai, bi) and np.array_equal(af.view(np.uint32), bf.view(np.uint32)), (g, side)
    ref.close()
    eng.load_weights(base)  # (c)
    st = eng.stats()
    assert B.FC0_FORMATS[int(st["fc0_format"])] == fmt0 and int(st["probe_outside"]) in (0, 1)
    p1, v1 = eng.evaluate_pv(x)
    assert np.array_equal(p0.view(np.uint32), p1.view(np.uint32)) and np.array_equal(v0.view(np.uint32), v1.view(np.uint32))
    eng.close()
