import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import omok_ai_amd as oa
from omok_ai_amd import binding as B
n, k = 15, 16
games = 40
tensors = oa.weights.init_random(n, seed=3)
eng = oa.Engine(board_size=n, games=games, max_nodes=512, max_tables=128, max_batch_k=k, seed=11, net_mode=B.NET_F16X3_F16)
eng.load_weights(tensors)
sp = oa.SelfPlay(eng)
sp.reset()
BLK = 512 * 16
for ply in range(2):
    for rnd in range(2):
        nreq = sp.round_generate(rnd, k, 0.25, 0.03)
        x = sp.round_inputs().copy()
        sp.round_eval()
        ra = eng.operand_rows(0, nreq).copy()
        sp.round_scatter()
        eng.evaluate_pv(x)
        rb = eng.operand_rows(0, nreq).copy()
        live = 16 * BLK  # bytes without the pad
        diff = ra[:, :live] != rb[:, :live]
        print(f"ply {ply} round {rnd}: rows with differing bytes {int(diff.any(axis=1).sum())} / {nreq}")
        if diff.any():
            r = int(np.nonzero(diff.any(axis=1))[0][0])
            d = diff[r].reshape(16, 2, 32, 8, 16)  # block (tile*2+q), part, pixel, piece, byte
            blocks = sorted(set(np.nonzero(d.any(axis=(1, 2, 3, 4)))[0].tolist()))
            parts = sorted(set(np.nonzero(d.any(axis=(0, 2, 3, 4)))[0].tolist()))
            print(f"  row {r}: differing blocks {blocks} parts {parts}")
            cnt = 0
            for b in blocks:
                for part in parts:
                    px = np.nonzero(d[b, part].any(axis=(1, 2)))[0]
                    if len(px):
                        pcs = sorted(set(np.nonzero(d[b, part].any(axis=(0, 2)))[0].tolist()))
                        print(f"    block {b} (tile {b // 2} q {b % 2}) part {part}: pixels {[int(b // 2 * 32 + p) for p in px][:40]} pieces {pcs}")
                        if cnt < 2:
                            p0 = int(px[0])
                            va = ra[r, :live].reshape(16, 2, 32, 8, 16)[b, part, p0].view(np.float16)
                            vb = rb[r, :live].reshape(16, 2, 32, 8, 16)[b, part, p0].view(np.float16)
                            print("      round :", va.reshape(-1)[:16])
                            print("      rowwise:", vb.reshape(-1)[:16])
                            cnt += 1
            stones = [int(i) // 2 for i in np.nonzero(x[r][: 2 * n * n])[0]]
            print(f"  row {r} stones {stones} turnplane {x[r][2*n*n]}")
    sp.sample_actions(1.0, 30)
    sp.mirror_generate(); sp.mirror_eval(); sp.mirror_apply()
eng.close()
