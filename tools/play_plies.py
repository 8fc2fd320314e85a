"""Plays the first plies of one self-play episode (default: configs[1]) and prints the category times: the workload of the rocprofv3 passes of
tools/pmc_kernel.sh and of A-B timings (OMOK_MI355X_LIB selects another build of the library).
usage: python tools/play_plies.py [board games sims k plies [net_mode [max_nodes]]]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import omok_ai_amd as oa
a = [int(x) for x in sys.argv[1:]]
n, games, sims, k, plies = (a + [15, 4096, 800, 16, 2][len(a):])[:5]
mode = a[5] if len(a) > 5 else 0
max_nodes = a[6] if len(a) > 6 else min(16384, 4 * sims + 1024)
eng = oa.Engine(board_size=n, games=games, max_nodes=max_nodes, max_tables=max(256, max_nodes // 4), max_batch_k=k, seed=0, net_mode=mode)
eng.load_random_weights(0)
sp = oa.SelfPlay(eng)
sp.reset(); sp.run(sims, k, max_plies=1)
sp.set_episode(1); sp.reset()
eng.set_profiling(1); eng.reset_stats()
t0 = time.perf_counter(); st = sp.run(sims, k, max_plies=plies); dt = time.perf_counter() - t0
print(json.dumps({"lib": os.path.basename(os.environ.get("OMOK_MI355X_LIB", "default")), "n": n, "max_nodes": max_nodes, "plies": plies, "seconds": round(dt, 4), "evals": st["evals"],
                  **{c: round(st[c], 2) for c in ("ms_round", "ms_tree", "ms_trunk", "ms_fc0", "ms_tail")}}), flush=True)
eng.close()
