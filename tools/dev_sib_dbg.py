import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import omok_ai_amd as oa
from omok_ai_amd import binding as B
n, k = 15, 16
games = int(sys.argv[1]) if len(sys.argv) > 1 else 40
mode = {"f16": B.NET_F16X3_F16, "fp6": B.NET_F16X3_FP6}[sys.argv[2] if len(sys.argv) > 2 else "f16"]
tensors = oa.weights.init_random(n, seed=3)
eng = oa.Engine(board_size=n, games=games, max_nodes=512, max_tables=128, max_batch_k=k, seed=11, net_mode=mode)
eng.load_weights(tensors)
ref = oa.Engine(board_size=n, games=64, max_nodes=8, max_tables=4, max_batch_k=k, net_mode=B.NET_F32)
ref.load_weights(tensors)
sp = oa.SelfPlay(eng)
sp.reset()
for ply in range(3):
    for rnd in range(3):
        nreq = sp.round_generate(rnd, k, 0.25, 0.03)
        x = sp.round_inputs().copy()
        p, v = sp.round_eval()
        p = np.array(p).reshape(nreq, -1).copy()
        v = np.array(v).reshape(-1).copy()
        sp.round_scatter()
        pp, vp = eng.evaluate_pv(x)
        pp = pp.reshape(nreq, -1)
        p32, v32 = ref.evaluate_pv(x)
        p32 = p32.reshape(nreq, -1)
        d = np.abs(p - pp).max(axis=1)
        d32 = np.abs(p - p32).max(axis=1)
        bad = np.nonzero(d > 1e-3)[0]
        print(f"ply {ply} round {rnd}: {nreq} rows, {int((d > 0).sum())} differ from row-by-row, {len(bad)} by > 1e-3; worst {d.max():.3e}; round vs f32 {d32.max():.2e}; rows vs f32 {np.abs(pp - p32).max():.2e}; "
              f"dv round {np.abs(v - v32.reshape(-1)).max():.2e} |v|<0.99: {int((np.abs(v32) < 0.99).sum())}")
        for r in bad[:12]:
            stones = [int(i) // 2 for i in np.nonzero(x[r][: 2 * n * n])[0]]
            first = (r // k) * k
            same_as_first = np.array_equal(x[r], x[first])
            newst = sorted(set(stones) - set(int(i) // 2 for i in np.nonzero(x[first][: 2 * n * n])[0])) if r != first else []
            print(f"    row {r} pos {r % k} d={d[r]:.2e} d32={d32[r]:.2e} nstones={len(stones)} new={newst} argmax_p={int(p[r].argmax())}/{int(pp[r].argmax())}")
    sp.sample_actions(1.0, 30)
    sp.mirror_generate()
    sp.mirror_eval()
    sp.mirror_apply()
eng.close(); ref.close()
