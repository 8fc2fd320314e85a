"""Phase profile of k_round (diagnostic build with -DKROUND_PROF: docs/experiments): plays PLIES plies of configs[1], then reads the per-launch, per-tree cycle records of the
LAST ply's 50 rounds and prints, per phase, the mean over the live trees and the breakdown of each round's SLOWEST tree (every tree's wave is resident at once: the slowest is the kernel).
usage: OMOK_MI355X_LIB=tools/ab/libomok_kprof.so python tools/kround_prof.py PLIES [PLIES ...]"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import omok_ai_amd as oa
from omok_ai_amd import binding as B
names = ["state+prev backups", "noise", "descents", "terminal backups", "leaf load", "picks", "win checks", "expand stores"]
lib = ctypes.CDLL(B.LIB_PATH)
eng = oa.Engine(board_size=15, games=4096, max_nodes=4224, max_tables=1056, max_batch_k=16, seed=0)
eng.load_random_weights(0)
sp = oa.SelfPlay(eng)
sp.set_episode(1); sp.reset()
done = 0
buf = np.zeros((64, 8192, 8), dtype=np.uint64)
for target in [int(x) for x in sys.argv[1:]]:
    sp.run(800, 16, max_plies=target - done)
    done = target
    lib.omok_debug_kround_prof(buf.ctypes.data_as(ctypes.c_void_p))
    side = (target - 1) & 1
    rec = buf[:50, side * 4096:(side + 1) * 4096, :].astype(np.float64)  # [round][game][phase]
    alive, _, _ = sp.game_info()
    tot = rec.sum(axis=2)
    live = tot[1] > 0
    print(f"== ply {target - 1}: {int(live.sum())} trees with records (alive now {int(alive.sum())}), cycles at the shader clock counter (100 MHz ticks x ?): raw units")
    mean = rec[1:, live, :].mean(axis=(0, 1))
    print("   mean over trees and rounds 1..49:  " + "  ".join(f"{n} {v:.0f}" for n, v in zip(names, mean)) + f"   | total {mean.sum():.0f}")
    worst = []
    for r in range(1, 50):
        g = int(np.argmax(tot[r]))
        worst.append(rec[r, g, :])
    worst = np.array(worst)
    wm = worst.mean(axis=0)
    print("   slowest tree of each round, mean:  " + "  ".join(f"{n} {v:.0f}" for n, v in zip(names, wm)) + f"   | total {wm.sum():.0f}")
    q = np.percentile(tot[1:, live], [50, 90, 99, 100])
    print(f"   total per tree and round: median {q[0]:.0f}, p90 {q[1]:.0f}, p99 {q[2]:.0f}, max {q[3]:.0f}")
