"""Per-ply wall time of one configs[1] episode against the number of live games (episode mode: omok_selfplay_run one ply at a time).
usage: python tools/ply_times.py [games sims k board]"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import omok_ai_amd as oa
games, sims, k, n = [int(x) for x in (sys.argv[1:5] if len(sys.argv) >= 5 else (4096, 800, 16, 15))]
eng = oa.Engine(board_size=n, games=games, max_nodes=4 * sims + 1024, max_tables=sims + 256, max_batch_k=k, seed=0)
eng.load_random_weights(0)
sp = oa.SelfPlay(eng)
sp.reset(); sp.run(sims, k, max_plies=2)
sp.set_episode(1); sp.reset()
rows = []
t_all = time.perf_counter()
while True:
    alive = sp.alive_count
    if alive == 0:
        break
    t0 = time.perf_counter()
    sp.run(sims, k, max_plies=1)
    rows.append((alive, (time.perf_counter() - t0) * 1e3))
total = time.perf_counter() - t_all
print(f"episode: {len(rows)} plies, {total:.3f} s, {games / total:.1f} games/s")
edges = [0, 64, 192, 256, 512, 1024, 2048, 3072, 4095, 1 << 30]
for lo, hi in zip(edges[:-1], edges[1:]):
    sel = [(a, ms) for a, ms in rows if lo < a <= hi]
    if sel:
        ms = sum(m for _, m in sel)
        print(f"live games ({lo}, {hi}]: {len(sel)} plies, {ms / 1e3:.3f} s ({100 * ms / 1e3 / total:.1f} %), {ms / len(sel) / (sims // k):.3f} ms per round, "
              f"{1e6 * ms / sum(a for a, _ in sel) / sims:.2f} ns per simulation")
