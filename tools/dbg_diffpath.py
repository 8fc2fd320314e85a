"""debug: outputs of real search rounds on the difference path (forced MIXED operand format: no probe, no fallback) against the oracle's forward.
usage: [OMOK_MI355X_LIB=...] python tools/dbg_diffpath.py [games]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import omok_ai_amd as oa
from omok_ai_amd import binding as B
from oracle import oracle as O
n, k = 15, 16
games = int(sys.argv[1]) if len(sys.argv) > 1 else 224
tensors = oa.weights.init_random(n, seed=0)
eng = oa.Engine(board_size=n, games=games, max_nodes=512, max_tables=128, max_batch_k=k, seed=3, net_mode=B.NET_F16X3_MIXED)
eng.load_weights(tensors)
sp = oa.SelfPlay(eng)
sp.reset()
net = O.Net(n, tensors)
for rnd in range(3):
    nreq = sp.round_generate(rnd, k)
    x = sp.round_inputs().copy()
    p, v = sp.round_eval()
    p = np.array(p).reshape(nreq, -1); v = np.array(v).reshape(-1)
    sel = np.arange(0, nreq, max(1, nreq // 96))[:96]
    pc, vc = net.forward(x[sel], threads=16)
    dp = np.abs(p[sel][:, :n * n] - pc); dv = np.abs(v[sel] - vc)
    st = eng.stats()
    print(f"round {rnd}: rows {nreq} children2 launches {st['children2_launches']} fc0_format {st['fc0_format']}  max|dp| {dp.max():.3e} max|dv| {dv.max():.3e}  rows with |dp| > 1e-3: {(dp.max(axis=1) > 1e-3).sum()} of {len(sel)}", flush=True)
    sp.round_scatter()
eng.close()
