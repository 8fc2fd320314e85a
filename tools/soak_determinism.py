"""Determinism soak at scale: the same episode (same seed) played on fresh engines must give bit-identical replay records.
A data race in a kernel (barriers, LDS-DMA ordering) shows up here as a mismatch.   usage: python tools/soak_determinism.py [games] [sims] [runs] [board] [K] [fp6|f16|auto]
Use sims > 225 (e.g. 1024 games x 512 sims): below that the root never becomes fully expanded, every simulation expands a random
untried root child and the replay does not depend on the net at all."""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import omok_ai_amd as oa

games = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
sims = int(sys.argv[2]) if len(sys.argv) > 2 else 128
runs = int(sys.argv[3]) if len(sys.argv) > 3 else 3
n = int(sys.argv[4]) if len(sys.argv) > 4 else 15
k = int(sys.argv[5]) if len(sys.argv) > 5 else 16
mode = {"auto": oa.binding.NET_F16X3, "fp6": oa.binding.NET_F16X3_FP6, "f16": oa.binding.NET_F16X3_F16}[sys.argv[6] if len(sys.argv) > 6 else "auto"]
digests = []
for r in range(runs):
    eng = oa.Engine(board_size=n, games=games, max_nodes=min(16384, 4 * sims + 1024), max_tables=max(256, (4 * sims + 1024) // 4),
                    max_batch_k=k, seed=123, net_mode=mode)
    eng.load_random_weights(0)
    sp = oa.SelfPlay(eng)
    sp.reset()
    st = sp.run(sims, k, 0.25, 0.03, 1.0, 30, 0)
    _, _, plies = sp.game_info()
    rec = sp.replay_record_bytes()
    total = int(plies.sum())
    buf = torch.zeros(total * rec, dtype=torch.uint8, device="cuda:0")
    got = sp.replay_pack_into(buf.data_ptr(), total)
    h = buf.cpu().numpy().reshape(total, rec)
    d = hashlib.sha256(h[np.lexsort(h[:, ::-1].T)].tobytes()).hexdigest()
    digests.append(d)
    print(f"run {r}: board {n} K {k} format {oa.binding.FC0_FORMATS[int(eng.stats()['fc0_format'])]} games {int(st['finished'])} plies {total} sims {int(st['sims'])} sha256 {d[:16]}", flush=True)
    eng.close()
assert len(set(digests)) == 1, "NONDETERMINISTIC: " + str(digests)
print("deterministic over", runs, "runs")
