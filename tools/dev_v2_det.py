"""Development check: two identical omok_selfplay_run runs on the difference path -> same packed replay bytes?  For both board sizes and both
operand formats (forced through net modes).  usage: python tools/dev_v2_det.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import omok_ai_amd as oa
from omok_ai_amd import binding as B

def run(n, games, count, k, plies, mode, cache=True):
    eng = oa.Engine(board_size=n, games=games, max_nodes=4 * count + 256, max_tables=count + 64, max_batch_k=k, seed=3, net_mode=mode)
    eng.load_weights(oa.weights.init_random(n, seed=0))
    eng.set_base_cache(cache)
    sp = oa.SelfPlay(eng); sp.reset(); sp.run(count, k, max_plies=plies)
    rec = sp.replay_record_bytes(); cap = games * plies
    buf = torch.zeros(cap * rec, dtype=torch.uint8, device="cuda")
    got = sp.replay_pack_into(buf.data_ptr(), cap)
    data = buf[: got * rec].cpu().numpy().copy()
    dumps = [sp.tree_dump(g, sd) for g in range(min(games, 32)) for sd in (0, 1)]
    eng.close()
    return data, dumps

def dump_diff(a, b):
    worst, trees = 0.0, 0
    for (ai, af), (bi, bf) in zip(a, b):
        if ai.shape != bi.shape or not np.array_equal(ai, bi):
            return "node records differ"
        neq = af.view(np.uint32) != bf.view(np.uint32)
        if neq.any():
            trees += 1
            worst = max(worst, float(np.abs(af[neq] - bf[neq]).max()))
    return f"{trees} trees differ, max |d| {worst:.2e}"

for n, games, count, k, plies in ((9, 256, 48, 8, 5), (15, 256, 96, 16, 4)):
    for name, mode in (("fp6", B.NET_F16X3_FP6), ("f16", B.NET_F16X3_F16)):
        a, da = run(n, games, count, k, plies, mode)
        res = []
        for rep in range(3):
            b, db = run(n, games, count, k, plies, mode)
            res.append((int((a != b).sum()), dump_diff(da, db)))
        c, dc = run(n, games, count, k, plies, mode, cache=False)
        print(f"n={n} {name}: 3 repeats (replay bytes differing, dumps) {res}; cache off {int((a != c).sum())} of {a.size}, {dump_diff(da, dc)}", flush=True)
