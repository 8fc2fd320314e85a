"""Two ranks of the REAL engine on the one GPU of the box (gloo): shard offsets, record counts and the order of the gathered replay
records.  What a multi-GPU run relies on: rank r owns games [r G, (r + 1) G), every game's RNG streams are keyed by its
GLOBAL id, the gather returns the ranks' records in rank order = global game order -- so two ranks with G games each must produce,
byte for byte, the packed records of ONE engine with 2 G games.  (RCCL itself needs one GPU per rank and is exercised by the driver's
multi-GPU bench; this test covers everything around it: launch, sharding, packing, counts exchange, uneven all-gather-v.)"""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import omok_ai_amd as oa
from omok_ai_amd import binding as B

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_on_one_gpu_equal_one_engine_with_all_the_games(tmp_path):
    import torch
    n, games, count, k, seed = 9, 24, 32, 8, 7
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "gathered.npz")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # (this pool's driver only supports dmabuf IPC: see DESIGN 7)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "tests", "rehearsal_worker.py"), out, str(n), str(games), str(count), str(k), str(seed)]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    got = np.load(out)
    assert int(got["world"]) == 2 and list(got["offsets"]) == [0, games]
    assert got["secs"] == 2.0 and got["finished"] == 2 * games  # max over ranks of the time, sum over ranks of the counters
    counts = got["counts"]
    assert len(counts) == 2 and counts.sum() == len(got["records"]) == got["plies"]
    assert np.array_equal(got["records"][: counts[0]], got["own"])  # rank 0's records come first

    eng = oa.Engine(board_size=n, games=2 * games, max_nodes=1024, max_tables=512, max_batch_k=k, seed=seed, game_offset=0, net_mode=B.NET_F32)
    eng.load_random_weights(0)
    sp = oa.SelfPlay(eng)
    sp.reset()
    sp.run(count, k)
    rec = sp.replay_record_bytes()
    buf = torch.empty(2 * games * n * n * rec, dtype=torch.uint8, device="cuda")
    cnt = sp.replay_pack_into(buf.data_ptr(), 2 * games * n * n)
    single = buf[: cnt * rec].view(cnt, rec).cpu().numpy()
    _, _, plies = sp.game_info()
    assert counts[0] == plies[:games].sum() and counts[1] == plies[games:].sum()  # (uneven: game lengths differ)
    assert np.array_equal(got["records"], single), "rank order is not global game order, or a game depends on its shard"
    eng.close()
