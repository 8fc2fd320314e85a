"""A-B timing of library builds inside one GPU session: for every libomok_*.so given, one process per (library, workload) that plays
the first plies of configs[1] and configs[2] with HIP-event profiling of every launch and prints the kernel-category times.
usage: python tools/ab_lib.py LIB [LIB ...]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time, json
sys.path.insert(0, %r)
import omok_ai_amd as oa
n, games, sims, k, plies = [int(x) for x in sys.argv[1:6]]
eng = oa.Engine(board_size=n, games=games, max_nodes=min(16384, 4 * sims + 1024), max_tables=max(256, (4 * sims + 1024) // 4), max_batch_k=k, seed=0)
eng.load_random_weights(0)
sp = oa.SelfPlay(eng)
sp.reset(); sp.run(sims, k, max_plies=1)
sp.set_episode(1); sp.reset()
eng.set_profiling(1); eng.reset_stats()
t0 = time.perf_counter(); st = sp.run(sims, k, max_plies=plies); dt = time.perf_counter() - t0
print(json.dumps({"n": n, "seconds": dt, "ms_round": st["ms_round"], "ms_tree": st["ms_tree"], "ms_trunk": st["ms_trunk"], "ms_fc0": st["ms_fc0"], "ms_tail": st["ms_tail"]}))
''' % ROOT
for lib in sys.argv[1:]:
    for cfg in [c for c in ((15, 4096, 800, 16, 3), (9, 16384, 200, 8, 4)) if str(c[0]) in os.environ.get("AB_BOARDS", "15,9").split(",")]:
        env = dict(os.environ, OMOK_MI355X_LIB=os.path.abspath(lib))
        out = subprocess.run([sys.executable, "-c", CHILD] + [str(x) for x in cfg], env=env, capture_output=True, text=True, timeout=300)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        print(os.path.basename(lib), line[-1] if line else out.stderr[-500:], flush=True)
