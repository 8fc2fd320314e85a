"""Development check of fc0's two operand formats (block-scaled fp6 / f16 correction terms) on the GPU box: evaluate_pv of the
forced formats against the fp32 kernels and the oracle, the commit-time probe of the automatic mode, commit time, and the sibling
paths (copy / difference) of search rounds in the f16 format.  Prints; asserts nothing.  `python tools/fc0_format_ab.py [quick]`."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import omok_ai_amd as oa  # noqa: E402
from omok_ai_amd import binding as B  # noqa: E402
from oracle import oracle as O  # noqa: E402


def random_positions(n, count, seed):
    rng = np.random.default_rng(seed)
    out = np.zeros((count, 3 * n * n), dtype=np.float32)
    for i in range(count):
        env = O.Environment(n)
        for c in rng.permutation(n * n)[: int(rng.integers(0, n * n - 1))]:
            env.place_stone(int(c))
        out[i] = env.encode_nn_input(int(rng.integers(0, 2)))
    return out


def rows_check(n, seed):
    tensors = oa.weights.init_random(n, seed=seed)
    x = random_positions(n, 300, 7)
    res = {}
    for name, mode in (("fp6", B.NET_F16X3_FP6), ("f16", B.NET_F16X3_F16), ("f32", B.NET_F32), ("auto", B.NET_F16X3)):
        eng = oa.Engine(board_size=n, games=64, max_nodes=16, max_tables=8, max_batch_k=16, net_mode=mode)
        t0 = time.time()
        eng.load_weights(tensors)
        dt = time.time() - t0
        p, v = eng.evaluate_pv(x)
        lg, vp = eng.evaluate_logits(x)
        st = eng.stats()
        res[name] = (p.reshape(len(x), -1), v.reshape(-1), lg, vp)
        print(f"  n={n} seed={seed} {name}: load+commit {dt:.2f}s format={B.FC0_FORMATS[int(st['fc0_format'])]} probe rows {st['probe_rows']:.0f} "
              f"fp6 {st['probe_dp_fp6']:.2e}/{st['probe_dv_fp6']:.2e} f16 {st['probe_dp_f16']:.2e}/{st['probe_dv_f16']:.2e} logit max {st['probe_logit_max']:.1f}", flush=True)
        eng.close()
    pc, vc = O.Net(n, tensors).forward(x[:128], threads=8)
    for name in ("fp6", "f16", "auto"):
        p, v, lg, vp = res[name]
        p32, v32, lg32, vp32 = res["f32"]
        print(f"  n={n} seed={seed} {name}: vs f32 |dp| {np.abs(p - p32).max():.2e} |dv| {np.abs(v - v32).max():.2e} |dlogit| {np.abs(lg - lg32).max():.2e} "
              f"|dvpre| {np.abs(vp - vp32).max():.2e};  vs oracle |dp| {np.abs(p[:128] - pc).max():.2e} |dv| {np.abs(v[:128] - vc).max():.2e}", flush=True)


def rounds_check(games, mode, name):
    n, k, count = 15, 16, 96
    tensors = oa.weights.init_random(n, seed=3)
    eng = oa.Engine(board_size=n, games=games, max_nodes=512, max_tables=128, max_batch_k=k, seed=11, net_mode=mode)
    eng.load_weights(tensors)
    ref = oa.Engine(board_size=n, games=games, max_nodes=8, max_tables=4, max_batch_k=k, net_mode=B.NET_F32)
    ref.load_weights(tensors)
    sp = oa.SelfPlay(eng)
    sp.reset()
    rows, dp, dv, dpp, dvp, differing = 0, 0.0, 0.0, 0.0, 0.0, 0
    for ply in range(3):
        for rnd in range(count // k):
            nreq = sp.round_generate(rnd, k, 0.25, 0.03)
            x = sp.round_inputs().copy()
            p, v = sp.round_eval()
            p, v = np.array(p).reshape(nreq, -1).copy(), np.array(v).reshape(-1).copy()
            sp.round_scatter()
            if rnd == 0:
                continue
            p32, v32 = ref.evaluate_pv(x)
            pp, vp = eng.evaluate_pv(x)
            p32, pp = p32.reshape(nreq, -1), pp.reshape(nreq, -1)
            dp, dv = max(dp, np.abs(p - p32).max()), max(dv, np.abs(v - v32.reshape(-1)).max())
            dpp, dvp = max(dpp, np.abs(p - pp).max()), max(dvp, np.abs(v - vp.reshape(-1)).max())
            differing += int((p.view(np.uint32) != pp.view(np.uint32)).any(axis=1).sum())
            rows += nreq
        sp.sample_actions(1.0, 30)
        sp.mirror_generate()
        sp.mirror_eval()
        sp.mirror_apply()
    print(f"  rounds {name} games={games}: {rows} rows: vs fp32 |dp| {dp:.2e} |dv| {dv:.2e}; vs row-by-row |dp| {dpp:.2e} |dv| {dvp:.2e}, {differing} rows differ", flush=True)
    eng.close()
    ref.close()


def timing(n, games, k, sims, plies):
    for name, mode in (("fp6", B.NET_F16X3_FP6), ("f16", B.NET_F16X3_F16)):
        eng = oa.Engine(board_size=n, games=games, max_nodes=4 * sims + 64, max_tables=sims + 64, max_batch_k=k, seed=0, net_mode=mode)
        eng.load_random_weights(0)
        sp = oa.SelfPlay(eng)
        sp.reset()
        sp.run(sims, k, max_plies=1)
        sp.reset()
        eng.set_profiling(1)
        eng.reset_stats()
        t0 = time.time()
        st = sp.run(sims, k, max_plies=plies)
        dt = time.time() - t0
        print(f"  timing n={n} {name}: {plies} plies of {games} games x {sims} sims in {dt:.3f}s; ms trunk {st['ms_trunk']:.0f} fc0 {st['ms_fc0']:.0f} tail {st['ms_tail']:.0f} "
              f"tree {st['ms_tree']:.0f}", flush=True)
        eng.close()


if __name__ == "__main__":
    quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
    for n in (9, 15):
        rows_check(n, 1)
    if not quick:
        rows_check(9, 0)
        rows_check(15, 0)
    rounds_check(40, B.NET_F16X3_F16, "f16 copy path")
    rounds_check(448, B.NET_F16X3_F16, "f16 difference path")
    rounds_check(448, B.NET_F16X3_FP6, "fp6 difference path")
    if not quick:
        timing(15, 4096, 16, 800, 2)
        timing(9, 16384, 8, 200, 3)
