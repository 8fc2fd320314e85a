"""Kernel-level timing of the net forward through the C ABI (evaluate_pv), HIP-event times.
usage: python tools/bench_net.py [batch] [reps]   (env OMOK_DBG_TRUNK = ablation bits, timing only)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import omok_ai_amd as oa

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
n = 15
eng = oa.Engine(board_size=n, games=B // 16, max_nodes=8, max_tables=4, max_batch_k=16)
eng.load_random_weights(0)
rng = np.random.default_rng(0)
x = (rng.random((B, 3 * n * n)) < 0.2).astype(np.float32)
eng.evaluate_pv(x[:4096])
eng.set_profiling(True)
eng.reset_stats()
for _ in range(reps):
    eng.evaluate_pv(x)
st = eng.stats()
print(f"B={B} dbg={os.environ.get('OMOK_DBG_TRUNK','0')}: trunk {st['ms_trunk']/reps:.3f} ms  fc0 {st['ms_fc0']/reps:.3f} ms  tail {st['ms_tail']/reps:.3f} ms")
