"""Multi-GPU plumbing: games shard by global id, one process per GPU, no hot-path collective.

The reference is single-process; self-play games are independent
(alpha-zero/src/parallel_mcts_executor.rs:200-205), so rank r simply owns games
[r*G, (r+1)*G) (engine `game_offset`), and the RNG streams are keyed by the global game id, which
makes results independent of the number of shards.  The only exchange is the optional episode-end
gather of replay tuples (s, pi, z).  Works with any torch.distributed backend ("nccl" = RCCL over
xGMI on the GPU box, "gloo" in the CPU tests).
"""
import os

import torch
import torch.distributed as dist


def shard_info():
    """(rank, local_rank, world_size) from the torchrun environment."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def game_offset(rank, games_per_rank):
    return rank * games_per_rank


def reduce_timing(seconds, counters, device):
    """max over ranks of the wall time, sum over ranks of the counters."""
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    c = torch.tensor(list(counters), dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
    return float(t[0]), [float(x) for x in c]


def gather_replay(records):
    """All-gather-v of packed replay records: `records` = this rank's LIVE records, uint8 [count, record_bytes] (counts differ
    between ranks).  Returns (all records [sum(counts), record_bytes] in rank order = global game order, counts list).

    Two steps: an all-gather of the world's counts (8 bytes per rank), then exact-size point-to-point transfers
    grouped into one batch -- on RCCL that is ncclGroupStart + one ncclSend / ncclRecv per peer, and on an 8-GPU xGMI node
    every peer pair has its own link, so the exchange is not ring-bound and ships only live bytes (a fixed-capacity slab
    all-gather moves ~4x more: a 15x15 game uses ~59 of its 225 record slots)."""
    rec = records.shape[1]
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return records, [int(records.shape[0])]
    world, rank = dist.get_world_size(), dist.get_rank()
    cnt = torch.tensor([records.shape[0]], dtype=torch.int64, device=records.device)
    counts_t = torch.zeros(world, dtype=torch.int64, device=records.device)
    dist.all_gather_into_tensor(counts_t, cnt)
    counts = [int(c) for c in counts_t.tolist()]
    offs = [0]
    for c in counts:
        offs.append(offs[-1] + c)
    out = torch.empty((offs[-1], rec), dtype=torch.uint8, device=records.device)
    mine = records.contiguous()
    out[offs[rank]:offs[rank + 1]] = mine
    ops = []
    for peer in range(world):
        if peer == rank:
            continue
        if counts[rank] > 0:
            ops.append(dist.P2POp(dist.isend, mine, peer))
        if counts[peer] > 0:
            dst = out[offs[peer]:offs[peer + 1]]  # whole rows of a contiguous matrix: a contiguous view (ncclRecv writes it in place)
            assert dst.is_contiguous()
            ops.append(dist.P2POp(dist.irecv, dst, peer))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return out, counts
