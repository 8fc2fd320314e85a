"""A/B of two builds of the library on the same inputs (bit-identity of kernel rewrites that must not change arithmetic).
usage: OMOK_MI355X_LIB=<lib.so> python tools/ab_net.py <out.npz> [n]   (run once per build, then compare the two files with
python tools/ab_net.py --compare a.npz b.npz)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

if sys.argv[1] == "--compare":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    ok = True
    for k in a.files:
        same = np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32))
        print(k, "bit-identical" if same else f"DIFFERENT: max abs diff {np.abs(a[k].astype(np.float64) - b[k].astype(np.float64)).max():.3e}")
        ok &= same
    sys.exit(0 if ok else 1)

import omok_ai_amd as oa
out = {}
for n in (9, 15):
    eng = oa.Engine(board_size=n, games=64, max_nodes=256, max_tables=64, max_batch_k=16, seed=1)
    eng.load_random_weights(0)
    rng = np.random.default_rng(0)
    hw = n * n
    x = np.zeros((700, 3 * hw), dtype=np.float32)
    for i in range(len(x)):  # encoder-layout rows of random positions
        k = int(rng.integers(0, hw))
        cells = rng.permutation(hw)[:k]
        turn = k % 2
        for j, c in enumerate(cells):
            mine = (j % 2) == turn
            x[i, 2 * c + (0 if mine else 1)] = 1.0
        x[i, 2 * hw:] = 1.0 if turn == 0 else 0.0
    p, v = eng.evaluate_pv(x)           # FROM_F32 path of the trunk
    lg, vp = eng.evaluate_logits(x)
    out[f"p{n}"], out[f"v{n}"], out[f"lg{n}"] = p, v, lg
    sp = oa.SelfPlay(eng)               # board-bits path of the trunk: a few plies of self-play, then the trees' statistics
    sp.reset()
    sp.run(32, 16, 0.25, 0.03, 1.0, 30, 6)
    out[f"w{n}"] = np.concatenate([sp.tree_dump(g, s)[1].reshape(-1) for g in range(8) for s in (0, 1)])
    eng.close()
np.savez(sys.argv[1], **out)
print("saved", sys.argv[1])
