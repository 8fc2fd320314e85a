"""Development check of the difference path's determinism at N = 9, f16 operand format: two engines in lockstep (base cache on / off), p / v of every
round compared bit for bit; prints the rows that differ.  usage: python tools/dev_v2_rows.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import omok_ai_amd as oa
from omok_ai_amd import binding as B

n, games, k, count = 9, 256, 8, 48
tensors = oa.weights.init_random(n, seed=0)
engs = []
for cache in (True, False, True):
    eng = oa.Engine(board_size=n, games=games, max_nodes=4 * count + 256, max_tables=count + 64, max_batch_k=k, seed=3, net_mode=B.NET_F16X3_F16)
    eng.load_weights(tensors)
    eng.set_base_cache(cache)
    sp = oa.SelfPlay(eng); sp.reset()
    engs.append((eng, sp))
for ply in range(5):
    for rnd in range(count // k):
        outs = []
        for eng, sp in engs:
            nreq = sp.round_generate(rnd, k, 0.25, 0.03)
            x = sp.round_inputs().copy()
            p, v = sp.round_eval()
            outs.append((nreq, x, np.array(p).reshape(nreq, -1).copy(), np.array(v).reshape(-1).copy()))
            sp.round_scatter()
        for j in (1, 2):
            assert outs[j][0] == outs[0][0] and np.array_equal(outs[j][1], outs[0][1]), "requests differ"
            dp = (outs[j][2].view(np.uint32) != outs[0][2].view(np.uint32)).any(axis=1) | (outs[j][3].view(np.uint32) != outs[0][3].view(np.uint32))
            if dp.any():
                rows = np.nonzero(dp)[0]
                print(f"ply {ply} round {rnd} engine {j} ({'cache off' if j == 1 else 'cache on'}): {len(rows)} of {outs[0][0]} rows differ: {rows[:12]} max|dp| "
                      f"{np.abs(outs[j][2][rows] - outs[0][2][rows]).max():.2e} |dv| {np.abs(outs[j][3][rows] - outs[0][3][rows]).max():.2e}", flush=True)
                for r in rows[:4]:
                    xr = outs[0][1][r].reshape(-1)
                    print("   row", r, "stones", int(xr[: 2 * n * n].sum()), "game", r // k)
    for eng, sp in engs:
        sp.sample_actions(1.0, 30)
        sp.advance()
print("done")
