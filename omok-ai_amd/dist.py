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


def gather_replay(records, count):
    """All-gather-v of packed replay records.

    records: uint8 tensor [cap, record_bytes] of which the first `count` rows are live (same cap on
    every rank).  Returns the list (one per rank) of live record tensors.  Two collectives: the
    counts, then fixed-capacity slabs (on an 8-GPU xGMI node every peer pair has its own link, so
    the slab all-gather is not ring-bound)."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return [records[:count]]
    world = dist.get_world_size()
    cnt = torch.tensor([count], dtype=torch.int64, device=records.device)
    counts = [torch.zeros_like(cnt) for _ in range(world)]
    dist.all_gather(counts, cnt)
    slabs = [torch.empty_like(records) for _ in range(world)]
    dist.all_gather(slabs, records.contiguous())
    return [s[: int(c[0])] for s, c in zip(slabs, counts)]
