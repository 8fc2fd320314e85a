"""Development check: the same round evaluated three times by one engine (base cache off / on) -> identical bits?  N = 9, f16 format."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import omok_ai_amd as oa
from omok_ai_amd import binding as B
n, games, k, count = 9, 256, 8, 48
mode = B.NET_F16X3_F16 if len(sys.argv) < 2 or sys.argv[1] == "f16" else B.NET_F16X3_FP6
for cache in (False, True):
    eng = oa.Engine(board_size=n, games=games, max_nodes=4 * count + 256, max_tables=count + 64, max_batch_k=k, seed=3, net_mode=mode)
    eng.load_weights(oa.weights.init_random(n, seed=0))
    eng.set_base_cache(cache)
    sp = oa.SelfPlay(eng); sp.reset()
    for rnd in range(3):
        nreq = sp.round_generate(rnd, k, 0.25, 0.03)
        outs = []
        for rep in range(4):
            p, v = sp.round_eval()
            outs.append((np.array(p).reshape(nreq, -1).copy(), np.array(v).reshape(-1).copy()))
        for rep in range(1, 4):
            dp = (outs[rep][0].view(np.uint32) != outs[0][0].view(np.uint32)).any(axis=1) | (outs[rep][1].view(np.uint32) != outs[0][1].view(np.uint32))
            print(f"cache {cache} round {rnd} repeat {rep}: {int(dp.sum())} of {nreq} rows differ from repeat 0", flush=True)
        sp.round_scatter()
    eng.close()
