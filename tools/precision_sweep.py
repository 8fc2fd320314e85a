"""Net parity sweep on the GPU box: GPU engine (split-operand MFMA path) in the automatic mode and in both forced operand formats of fc0
(block-scaled fp6 / f16 correction terms) vs the fp32 CPU oracle over several weight seeds and many random positions; prints the
worst |dp| / |dv| per (N, seed, mode), the format the commit probe chose and the probe's own figures.
usage: python tools/precision_sweep.py [positions] [seeds]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import omok_ai_amd as oa
from omok_ai_amd import binding as B
from oracle import oracle as O

count = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
worst = {"auto": 0.0, "fp6": 0.0, "f16": 0.0}
for n in (9, 15):
    rng = np.random.default_rng(100 + n)
    xs = []
    for i in range(count):
        env = O.Environment(n)
        stones = int(rng.integers(0, 6)) if i % 3 == 0 else int(rng.integers(0, n * n - 1))  # a third of the positions nearly empty (early plies)
        for c in rng.permutation(n * n)[:stones]:
            env.place_stone(int(c))
        xs.append(env.encode_nn_input(int(rng.integers(0, 2))))
    x = np.stack(xs)
    for seed in range(seeds):
        tensors = oa.weights.init_random(n, seed=seed)
        pc, vc = O.Net(n, tensors).forward(x, threads=os.cpu_count() or 8)
        line = f"N={n} seed={seed}: {count} positions"
        for name, mode in (("auto", B.NET_F16X3), ("fp6", B.NET_F16X3_FP6), ("f16", B.NET_F16X3_F16)):
            eng = oa.Engine(board_size=n, games=128, max_nodes=8, max_tables=4, max_batch_k=16, net_mode=mode)
            eng.load_weights(tensors)
            st = eng.stats()
            p, v = eng.evaluate_pv(x)
            dp = float(np.abs(p.reshape(count, -1) - pc.reshape(count, -1)).max())
            dv = float(np.abs(v.ravel() - vc.ravel()).max())
            worst[name] = max(worst[name], dp, dv)
            line += f"  | {name}: max|dp|={dp:.2e} max|dv|={dv:.2e}"
            if name == "auto":
                line += f" (chose {B.FC0_FORMATS[int(st['fc0_format'])]}; probe fp6 {st['probe_dp_fp6']:.1e}/{st['probe_dv_fp6']:.1e} f16 {st['probe_dp_f16']:.1e}/{st['probe_dv_f16']:.1e})"
            eng.close()
        print(line, flush=True)
print("worst over all nets: " + ", ".join(f"{k} {v:.2e}" for k, v in worst.items()) + "  (contract 1e-3)")
