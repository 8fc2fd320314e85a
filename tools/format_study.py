"""fc0 operand formats on the path the search rounds take (sibling base + window differences), against the ORACLE's fp32 forward: for every (board, weight seed) and
format in {fp6, mixed, f16} plays two plies of step-wise rounds on the difference path and reports the worst |dp|, |dv| of the rounds' outputs and of the same rows
evaluated one by one (omok_evaluate_pv: full rows only).  usage: python tools/format_study.py [rows_per_round_checked]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import omok_ai_amd as oa
from omok_ai_amd import binding as B
from oracle import oracle as O

per = int(sys.argv[1]) if len(sys.argv) > 1 else 64
MODES = {"fp6": B.NET_F16X3_FP6, "mixed": B.NET_F16X3_MIXED, "f16": B.NET_F16X3_F16}
for n, games, k, seeds in ((9, 160, 8, (0, 1, 2, 3)), (15, 224, 16, (0, 1, 2, 3))):
    for seed in seeds:
        tensors = oa.weights.init_random(n, seed=seed)
        net = O.Net(n, tensors)
        line = f"N={n} seed={seed}:"
        for tag, mode in MODES.items():
            eng = oa.Engine(board_size=n, games=games, max_nodes=512, max_tables=128, max_batch_k=k, seed=21, net_mode=mode)
            eng.load_weights(tensors)
            sp = oa.SelfPlay(eng)
            sp.reset()
            rng = np.random.default_rng(0)
            rows, dp, dv, dpr, dvr = 0, 0.0, 0.0, 0.0, 0.0
            for ply in range(2):
                for rnd in range(5):
                    nreq = sp.round_generate(rnd, k, 0.25, 0.03)
                    x = sp.round_inputs().copy()
                    p, v = sp.round_eval()
                    p, v = np.array(p).reshape(nreq, -1).copy(), np.array(v).reshape(-1).copy()
                    sp.round_scatter()
                    if rnd == 0:
                        continue
                    pick = rng.choice(nreq, size=per, replace=False)
                    pc, vc = net.forward(x[pick], threads=16)
                    dp, dv = max(dp, float(np.abs(p[pick] - pc).max())), max(dv, float(np.abs(v[pick] - vc).max()))
                    pp, vp = eng.evaluate_pv(x[pick])
                    dpr, dvr = max(dpr, float(np.abs(pp.reshape(len(pick), -1) - pc).max())), max(dvr, float(np.abs(vp.reshape(-1) - vc).max()))
                    rows += len(pick)
                sp.sample_actions(1.0, 30)
                sp.advance()
            fmt = B.FC0_FORMATS[int(eng.stats()["fc0_format"])]
            eng.close()
            line += f"  | {tag} ({fmt}): rounds {dp:.2e}/{dv:.2e} rows {dpr:.2e}/{dvr:.2e}"
        print(line + f"  ({rows} rows)", flush=True)
