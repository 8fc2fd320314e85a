"""Experiment: L engines of G / L games each (global game ids by game_offset), driven from L host threads on L streams of ONE GPU,
against one engine of G games.  python tools/exp_lanes.py [G] [sims] [lanes...]"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import omok_ai_amd as oa

G = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
sims = int(sys.argv[2]) if len(sys.argv) > 2 else 800
lanes_list = [int(x) for x in sys.argv[3:]] or [1, 2, 4]
n, k = 15, 16
tensors = oa.weights.init_random(n, seed=0)
for L in lanes_list:
    g = G // L
    engs = []
    for i in range(L):
        e = oa.Engine(board_size=n, games=g, max_nodes=min(16384, 4 * sims + 1024), max_tables=max(256, (4 * sims + 1024) // 4), max_batch_k=k, seed=0, game_offset=i * g)
        e.load_weights(tensors)
        engs.append((e, oa.SelfPlay(e)))
    def ply4(sp):
        sp.reset(); sp.run(sims, k, max_plies=3)
    for e, sp in engs:
        ply4(sp)
    for e, sp in engs:
        sp.set_episode(1); sp.reset()
    res = [None] * L
    def work(i):
        res[i] = engs[i][1].run(sims, k)
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(i,)) for i in range(L)]
    for t in th: t.start()
    for t in th: t.join()
    dt = time.perf_counter() - t0
    fin = sum(r["finished"] for r in res)
    plies = np.concatenate([sp.game_info()[2] for e, sp in engs])
    print(f"lanes {L}: {fin:.0f} games in {dt:.3f} s = {fin / dt:.1f} games/s; mean plies {plies.mean():.2f}", flush=True)
    for e, sp in engs:
        e.close()
