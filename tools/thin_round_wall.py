"""tools/thin_round_wall.py G...: wall time per search round of engines with few games, HIP-event profiling OFF (no event records in the stream): against the sums of the
kernels' durations from tools/thin_round_gaps.sh this is what the launch boundaries of a small round cost."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import omok_ai_amd as oa
for games in [int(x) for x in sys.argv[1:]]:
    eng = oa.Engine(board_size=15, games=games, max_nodes=4224, max_tables=1056, max_batch_k=16, seed=0)
    eng.load_random_weights(0)
    sp = oa.SelfPlay(eng)
    sp.reset(); sp.run(800, 16, max_plies=1)
    sp.set_episode(1); sp.reset()
    eng.set_profiling(0)
    t0 = time.perf_counter(); sp.run(800, 16, max_plies=4); dt = time.perf_counter() - t0
    print(f"G={games}: 4 plies = 200 rounds in {dt * 1e3:.1f} ms: {dt / 200 * 1e6:.1f} us per round (ply-level kernels and one status read-back per ply included)", flush=True)
    eng.close()
