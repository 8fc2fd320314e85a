"""Worker of tests/test_gpu_rehearsal.py (not a test): one rank of a 2-rank run of the REAL engine, both ranks on the one GPU of the box,
`gloo` for the exchange (RCCL needs one GPU per rank).  Each rank plays its shard of the games (global ids rank * G .. rank * G + G - 1)
to the end, packs its replay records on the device and takes part in the all-gather-v of omok-ai_amd/dist.py; rank 0 saves what it
gathered.  Usage: python -m torch.distributed.run --nproc-per-node 2 ... tests/rehearsal_worker.py OUT.npz N GAMES COUNT K SEED"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import torch
    import torch.distributed as dist
    import omok_ai_amd as oa
    out, n, games, count, k, seed = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
    dist.init_process_group("gloo")
    rank, _, world = oa.dist.shard_info()
    offset = oa.dist.game_offset(rank, games)
    eng = oa.Engine(board_size=n, games=games, max_nodes=1024, max_tables=512, max_batch_k=k, device=0, seed=seed, game_offset=offset,
                    net_mode=oa.binding.NET_F32)  # (fp32 kernels: a row's bits do not depend on the batch it is evaluated in)
    eng.load_random_weights(0)
    sp = oa.SelfPlay(eng)
    sp.reset()
    st = sp.run(count, k)
    rec = sp.replay_record_bytes()
    buf = torch.empty(games * n * n * rec, dtype=torch.uint8, device="cuda:0")
    cnt = sp.replay_pack_into(buf.data_ptr(), games * n * n)
    live = buf[: cnt * rec].view(cnt, rec).cpu()
    allrec, counts = oa.dist.gather_replay(live)
    secs, (finished, plies) = oa.dist.reduce_timing(1.0 + rank, [st["finished"], st["ply_games"]], "cpu")
    if rank == 0:
        np.savez(out, records=allrec.numpy(), counts=np.array(counts), own=live.numpy(), secs=secs, finished=finished, plies=plies, world=world,
                 offsets=np.array([oa.dist.game_offset(r, games) for r in range(world)]))
    dist.barrier()
    dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
