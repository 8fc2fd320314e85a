"""world_size-2 gloo test (CPU) of the multi-GPU path: shard assignment by global game id, the
timing/counter reduction and the episode-end gather of replay tuples.  The CPU oracle stands in
for the engine; the property checked is the one the GPU engine relies on: a game's result depends
only on (seed, global game id), not on how games are sharded over ranks."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N, GAMES_PER_RANK, COUNT, K, PLIES = 9, 2, 16, 8, 6
REC = N * N + 3 + 4 * N * N + 4  # board, turn, pad to 4 | pi | z   (omok_replay_record_bytes layout)


def _play(n, games, offset):
    sys.path.insert(0, ROOT)
    from oracle import oracle as O
    hw = n * n
    sp = O.SelfPlay(n, games, cap_nodes=512, cap_tables=256, seed=5, game_offset=offset)
    sp.reset(np.full(hw, 1.0 / hw, dtype=np.float32))
    rng_p = np.full((games * K, hw), 1.0 / hw, dtype=np.float32)
    for _ in range(PLIES):
        for rnd in range(COUNT // K):
            inp = sp.round_generate(rnd, K, 0.25, 0.03)
            # deterministic stand-in evaluator: value from a hash of the input, uniform policy
            v = np.tanh(inp.reshape(len(inp), -1)[:, ::7].sum(axis=1) * 0.01).astype(np.float32)
            sp.round_scatter(rng_p[: len(inp)], v)
        sp.sample(1.0, 30)
        m = sp.mirror_generate()
        sp.advance(rng_p[: len(m)])
    recs = []
    for g in range(games):
        boards, turns, pi, z = sp.replay(g)
        for i in range(len(boards)):
            r = np.zeros(REC, dtype=np.uint8)
            r[:hw] = boards[i]
            r[hw] = turns[i]
            r[hw + 3: hw + 3 + 4 * hw] = pi[i].view(np.uint8)
            r[hw + 3 + 4 * hw:] = np.array([z[i]], dtype=np.float32).view(np.uint8)
            recs.append(r)
    return np.stack(recs)


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                      WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import omok_ai_amd as oa
    dist.init_process_group("gloo", rank=rank, world_size=world)
    r, lr, w = oa.dist.shard_info()
    assert (r, w) == (rank, world)
    recs = _play(N, GAMES_PER_RANK, oa.dist.game_offset(r, GAMES_PER_RANK))
    cap = GAMES_PER_RANK * N * N
    slab = torch.zeros((cap, REC), dtype=torch.uint8)
    slab[: len(recs)] = torch.from_numpy(recs)
    parts = oa.dist.gather_replay(slab, len(recs))
    secs, (games, plies) = oa.dist.reduce_timing(1.0 + rank, [GAMES_PER_RANK, len(recs)], "cpu")
    if rank == 0:
        np.savez(out, all=torch.cat(parts).numpy(), secs=secs, games=games, plies=plies,
                 sizes=np.array([len(p) for p in parts]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_matches_single_process(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "gathered.npz")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = np.load(out)
    single = _play(N, 2 * GAMES_PER_RANK, 0)  # the same four games in one process
    assert got["secs"] == 2.0 and got["games"] == 2 * GAMES_PER_RANK  # max of times, sum of counters
    assert got["plies"] == len(single) == got["sizes"].sum()
    assert np.array_equal(got["all"], single)  # rank-major gather == global game order, bit for bit
