"""Development check: the two children kernels of the difference path on full-size rounds (4096 games x K = 16 at N = 15, 16384 x 8 at N = 9): max |dp|, |dv| and rows that differ in bits."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import omok_ai_amd as oa
for n, games, k in ((15, 4096, 16), (9, 16384, 8)):
    tensors = oa.weights.init_random(n, seed=0)
    outs = []
    for which in (2, 1):
        eng = oa.Engine(board_size=n, games=games, max_nodes=512, max_tables=128, max_batch_k=k, seed=5)
        eng.load_weights(tensors)
        eng.set_children_kernel(which)
        sp = oa.SelfPlay(eng); sp.reset()
        per = []
        for rnd in range(6):
            nreq = sp.round_generate(rnd, k, 0.25, 0.03)
            p, v = sp.round_eval()
            per.append((np.array(p).reshape(nreq, -1).copy(), np.array(v).reshape(-1).copy()))
            sp.round_scatter()
        st = eng.stats()
        outs.append(per); eng.close()
        print(f"n={n} kernel {which}: launches children2 {st['children2_launches']:.0f} children1 {st['children1_launches']:.0f}", flush=True)
    for rnd in range(6):
        (pa, va), (pb, vb) = outs[0][rnd], outs[1][rnd]
        diff = int(((pa.view(np.uint32) != pb.view(np.uint32)).any(axis=1) | (va.view(np.uint32) != vb.view(np.uint32))).sum())
        print(f"n={n} round {rnd}: {len(pa)} rows, {diff} differ in bits, max|dp| {np.abs(pa - pb).max():.2e} max|dv| {np.abs(va - vb).max():.2e}", flush=True)
