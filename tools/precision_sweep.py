"""Net parity sweep on the GPU box: GPU engine (split-operand MFMA path) vs the fp32 CPU oracle over several weight seeds
and many random positions; prints the worst |dp| / |dv| per (N, seed).  usage: python tools/precision_sweep.py [positions]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import omok_ai_amd as oa
from oracle import oracle as O

count = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
worst = 0.0
for n in (9, 15):
    rng = np.random.default_rng(100 + n)
    xs = []
    for _ in range(count):
        env = O.Environment(n)
        for c in rng.permutation(n * n)[: int(rng.integers(0, n * n - 1))]:
            env.place_stone(int(c))
        xs.append(env.encode_nn_input(int(rng.integers(0, 2))))
    x = np.stack(xs)
    for seed in range(4):
        tensors = oa.weights.init_random(n, seed=seed)
        eng = oa.Engine(board_size=n, games=128, max_nodes=8, max_tables=4, max_batch_k=16)
        eng.load_weights(tensors)
        p, v = eng.evaluate_pv(x)
        pc, vc = O.Net(n, tensors).forward(x, threads=os.cpu_count() or 8)
        dp = float(np.abs(p.reshape(count, -1) - pc.reshape(count, -1)).max())
        dv = float(np.abs(v.ravel() - vc.ravel()).max())
        worst = max(worst, dp, dv)
        print(f"N={n} seed={seed}: {count} positions  max|dp|={dp:.3e}  max|dv|={dv:.3e}", flush=True)
        eng.close()
print(f"worst {worst:.3e}  (contract 1e-3)")
