"""Phase profile of the children kernels of the difference path (OMOK_SIB_PROF=1: k_sib_children, 2: k_sib_children2): plays the first plies of
configs[1] and lets the library print its in-kernel cycle counters to stderr.  usage: OMOK_SIB_PROF=2 python tools/sib_phase_profile.py [plies]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import omok_ai_amd as oa
plies = int(sys.argv[1]) if len(sys.argv) > 1 else 5
eng = oa.Engine(board_size=15, games=4096, max_nodes=4 * 800 + 1024, max_tables=1056, max_batch_k=16, seed=0)
eng.load_random_weights(0)
sp = oa.SelfPlay(eng)
sp.reset()
t0 = time.perf_counter()
sp.run(800, 16, max_plies=plies)
print(f"{plies} plies in {time.perf_counter() - t0:.3f} s", flush=True)
