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
    allrec, counts = oa.dist.gather_replay(torch.from_numpy(recs))  # counts exchange + exact-size all-gather-v
    secs, (games, plies) = oa.dist.reduce_timing(1.0 + rank, [GAMES_PER_RANK, len(recs)], "cpu")
    if rank == 0:
        np.savez(out, all=allrec.numpy(), secs=secs, games=games, plies=plies, sizes=np.array(counts))
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


# ---- the replay gather at the node's full width: 8 ranks, one of them without a single record --------------------------------
COUNTS8 = [3, 0, 5, 1, 7, 2, 4, 6]  # live records per rank (rank 1: a shard whose games recorded nothing)


def _records8(rank):
    """records whose bytes name their origin: [rank, index, rank ^ index, 0...] + a running pattern"""
    c = COUNTS8[rank]
    r = np.zeros((c, REC), dtype=np.uint8)
    for i in range(c):
        r[i, :3] = (rank, i, rank ^ i)
        r[i, 3:] = (np.arange(REC - 3) * (rank + 1) + i) % 251
    return r


def _worker8(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import omok_ai_amd as oa
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    posted = []
    real = dist.batch_isend_irecv

    def counting(ops):  # (the grouped exchange: ncclGroupStart + one ncclSend / ncclRecv per peer on RCCL)
        posted.append(len(ops))
        return real(ops)

    dist.batch_isend_irecv = counting
    allrec, counts = oa.dist.gather_replay(torch.from_numpy(_records8(rank)))
    dist.batch_isend_irecv = real
    assert counts == COUNTS8
    # sends to the 7 peers unless this rank has nothing, receives from every peer that has something
    expect = (7 if COUNTS8[rank] else 0) + sum(1 for p in range(world) if p != rank and COUNTS8[p] > 0)
    assert posted == [expect], (rank, posted, expect)
    secs, (games, recs) = oa.dist.reduce_timing(float(rank), [1, COUNTS8[rank]], "cpu")
    assert secs == 7.0 and games == 8 and recs == sum(COUNTS8)
    np.save(f"{out}.rank{rank}.npy", allrec.numpy())  # every rank holds the whole gather
    dist.barrier()
    dist.destroy_process_group()


def test_eight_rank_gather_replay_with_an_empty_rank(tmp_path):
    """configs[3] / [4]'s width on the CPU: 8 ranks, uneven counts, rank 1 empty: every rank ends up with all the records in rank
    order (= global game order), and a rank posts exactly one send per peer and one receive per non-empty peer (14 at most)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "g8")
    mp.spawn(_worker8, args=(8, port, out), nprocs=8, join=True)
    want = np.concatenate([_records8(r) for r in range(8)])
    assert want.shape[0] == sum(COUNTS8)
    for r in range(8):
        assert np.array_equal(np.load(f"{out}.rank{r}.npy"), want), r


# ---- data-parallel training step: gradients averaged over ranks (omok-ai_amd/train.py) ------------------------------
def _train_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import omok_ai_amd as oa
    from omok_ai_amd import train as T
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tensors = [np.asarray(t, np.float64) * 0.25 for t in oa.weights.init_random(N, seed=2)]
    x, pi, z = _train_batch(8)
    ph = T.TrainPhase(N, tensors, "cpu", dtype=torch.float64, allow_cpu=True)
    sl = slice(4 * rank, 4 * rank + 4)  # each rank trains on its half of the batch
    for _ in range(2):
        ph.step(torch.as_tensor(x[sl]), torch.as_tensor(pi[sl]), torch.as_tensor(z[sl]))
    if rank == 0:
        np.savez(out, *[p.detach().numpy() for p in ph.net.vars])
    flat = torch.cat([p.detach().reshape(-1) for p in ph.net.vars])
    both = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(both, flat)
    assert torch.equal(both[0], both[1])  # replicas stay bit-identical
    dist.barrier()
    dist.destroy_process_group()


def _train_batch(b):
    rng = np.random.default_rng(7)
    hw = N * N
    x = (rng.random((b, N, N, 3)) < 0.3).astype(np.float64)
    pi = rng.random((b, hw))
    pi /= pi.sum(axis=1, keepdims=True)
    z = rng.choice([-1.0, 0.0, 1.0], size=(b, 1))
    return x, pi, z


def test_two_rank_training_step_equals_single_process_on_the_union(tmp_path):
    """mean over ranks of the per-rank mean-loss gradients == gradient of the mean loss over the union (equal shares)."""
    sys.path.insert(0, ROOT)
    import omok_ai_amd as oa
    from omok_ai_amd import train as T
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "trained.npz")
    mp.spawn(_train_worker, args=(2, port, out), nprocs=2, join=True)
    got = np.load(out)
    tensors = [np.asarray(t, np.float64) * 0.25 for t in oa.weights.init_random(N, seed=2)]
    x, pi, z = _train_batch(8)
    ph = T.TrainPhase(N, tensors, "cpu", dtype=torch.float64, allow_cpu=True)
    for _ in range(2):
        ph.step(torch.as_tensor(x), torch.as_tensor(pi), torch.as_tensor(z))
    for i, p in enumerate(ph.net.vars):
        assert np.abs(got[f"arr_{i}"] - p.detach().numpy()).max() < 1e-12, i
