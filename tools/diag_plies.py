"""Per-ply counters of k_round (root descents, terminal simulations, multi-parent rounds) on a DIAGNOSTIC build of the library: apply
docs/experiments/r05_lds_staged_path.diff's -DKROUND_COUNT hunks (tree_kernels.hip, engine.cpp), build it as tools/ab/libomok_cnt.so and run
   OMOK_MI355X_LIB=tools/ab/libomok_cnt.so python3 tools/diag_plies.py
The product library prints nothing here (the counters do not exist in it)."""
import os, sys
sys.path.insert(0, "/root/repo")
import omok_ai_amd as oa
eng = oa.Engine(board_size=15, games=4096, max_nodes=4224, max_tables=1056, max_batch_k=16, seed=0)
eng.load_random_weights(0)
sp = oa.SelfPlay(eng)
sp.set_episode(1); sp.reset()
ply = 0
while sp.alive_count > 0:
    a = sp.alive_count
    eng.reset_stats()
    sp.run(800, 16, max_plies=1)
    sys.stderr.write(f"ply {ply} alive {a} ")
    sys.stderr.flush()
    eng.stats()
    ply += 1
