#!/usr/bin/env python3
"""bench.py — self-play throughput of the MI355X engine on BASELINE.json's metric.

A "step" is ONE self-play episode: `games` concurrent 15x15 games per GPU, two agents (trees) per
game, `sims` PUCT simulations per move in rounds of K with one batched net forward per round,
random-init net (seed 0), played until every game has ended (src/trainer.rs:95-205).
`value` = completed games / second over the timed steps, whole job (all ranks).

Default workload = BASELINE.json configs[1]: 4096 concurrent 15x15 games, 800 sims/move, K=16.
Multi-GPU (configs[3]): one process per GPU (torchrun), games sharded by global id
(game_offset = rank * games), no collective on the hot path; --gather adds the optional RCCL
all-gather of (s, pi, z) replay tuples at episode end (configs[4]).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

F16_DENSE_PEAK_TFLOPS = 2500.0   # MI355X_MICROARCH.md: bf16/f16 MFMA dense
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E spec
# HBM bytes per fc0 row from the committed rocprofv3 --pmc passes (profiles/README.md, B = 65536): FETCH_SIZE raw x 2
# (gfx950 correction for 128-B requests) minus the residual part that is fetched in exact 64-B requests, plus WRITE_SIZE.
FC0_HBM_BYTES_PER_ROW = {15: (3.873e6 * 1024 * 2 - 65536 * 28800.0 + 1.347e5 * 1024) / 65536}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=0)
    ap.add_argument("--board", type=int, default=15)
    ap.add_argument("--games", type=int, default=4096, help="concurrent games per GPU")
    ap.add_argument("--sims", type=int, default=800)
    ap.add_argument("--batch-k", type=int, default=16, help="evaluate_batch_size (src/config.rs:92)")
    ap.add_argument("--max-plies", type=int, default=0, help="debug: stop every episode after this many plies")
    ap.add_argument("--max-nodes", type=int, default=0)
    ap.add_argument("--max-tables", type=int, default=0)
    ap.add_argument("--net-mode", default="f16x3", choices=["f16x3", "f32"])
    ap.add_argument("--gather", action="store_true", help="RCCL all-gather of replay tuples at episode end")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="budget of the cpu_baseline leg (0 = skip)")
    ap.add_argument("--train-steps", type=int, default=20, help="training steps timed after the episode (0 = skip; batch 128)")
    ap.add_argument("--seed", type=int, default=0)
    return ap.parse_args()


def cpu_baseline(args, mean_plies):
    """The oracle (CPU restatement, kind 'port') timed on this host on a bounded sample of the same
    workload: G games x `sims` sims x a few plies, all host cores (OpenMP over net batches)."""
    from oracle import oracle as O
    import omok_ai_amd as oa
    cores = os.cpu_count() or 1
    n = args.board
    tensors = oa.weights.init_random(n, seed=0)
    net = O.Net(n, tensors)
    root_p, _ = net.forward(O.Environment(n).encode_nn_input(0)[None])
    games = max(1, cores // 2)
    # calibrate: one round of K sims
    sp = O.SelfPlay(n, games, cap_nodes=max(2048, args.sims * 2 + 64), cap_tables=1024, seed=args.seed)
    sp.reset(root_p[0])
    t0 = time.perf_counter()
    err, st = sp.run(net, args.batch_k, args.batch_k, max_plies=1, threads=cores)
    t_round = max(time.perf_counter() - t0, 1e-3)
    rounds_per_ply = (args.sims + args.batch_k - 1) // args.batch_k
    plies = int(max(1, min(4, args.cpu_seconds / (t_round * rounds_per_ply))))
    sims = args.sims
    if t_round * rounds_per_ply > 1.5 * args.cpu_seconds:  # even one ply is over budget: scale sims down, say so
        sims = max(args.batch_k, int(args.sims * args.cpu_seconds / (t_round * rounds_per_ply)) // args.batch_k * args.batch_k)
    sp = O.SelfPlay(n, games, cap_nodes=max(2048, args.sims * 2 + 64), cap_tables=1024, seed=args.seed)
    sp.reset(root_p[0])
    t0 = time.perf_counter()
    err, st = sp.run(net, sims, args.batch_k, max_plies=plies, threads=cores)
    dt = time.perf_counter() - t0
    assert err == 0
    sims_per_s = st["sims"] / dt
    games_per_s = sims_per_s / (args.sims * mean_plies)
    return {
        "value": games_per_s, "unit": "games/s", "cores": cores, "kind": "port",
        "sample": f"{games} games x {plies} plies x {sims} sims/move (K={args.batch_k}), {n}x{n}, oracle C restatement "
                  f"with OpenMP net; {sims_per_s:.1f} sims/s measured, converted with {mean_plies:.1f} plies/game from the GPU run",
        "sims_per_s": sims_per_s, "seconds": dt, "net_seconds": st["t_net"],
    }


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit(f"--gpus {args.gpus} needs torchrun --nproc-per-node {args.gpus}")
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import omok_ai_amd as oa
    from omok_ai_amd import binding as B

    n, games, k = args.board, args.games, args.batch_k
    max_nodes = args.max_nodes or min(16384, 4 * args.sims + 1024)
    max_tables = args.max_tables or max(256, max_nodes // 4)
    eng = oa.Engine(board_size=n, games=games, max_nodes=max_nodes, max_tables=max_tables, max_batch_k=k,
                    device=local_rank, net_mode=B.NET_F16X3 if args.net_mode == "f16x3" else B.NET_F32,
                    seed=args.seed, game_offset=rank * games)
    eng.load_random_weights(0)
    sp = oa.SelfPlay(eng)
    eng.set_profiling(True)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    gather_buf = None
    if args.gather:
        rec = sp.replay_record_bytes()
        cap = games * n * n
        gather_buf = torch.empty(cap * rec, dtype=torch.uint8, device=f"cuda:{local_rank}")

    def episode():
        sp.reset()
        st = sp.run(args.sims, k, 0.25, 0.03, 1.0, 30, args.max_plies)
        if args.gather:
            cnt = sp.replay_pack_into(gather_buf.data_ptr(), gather_buf.numel() // sp.replay_record_bytes())
            if world > 1:
                counts = [torch.zeros(1, dtype=torch.int64, device=gather_buf.device) for _ in range(world)]
                dist.all_gather(counts, torch.tensor([cnt], dtype=torch.int64, device=gather_buf.device))
                bufs = [torch.empty_like(gather_buf) for _ in range(world)]
                dist.all_gather(bufs, gather_buf)  # fixed-capacity slabs; counts say how much of each is live
        return st

    for _ in range(args.warmup):
        episode()
    eng.reset_stats()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        episode()
    barrier()
    dt = time.perf_counter() - t0
    st = eng.stats()
    alive, status, plies = sp.game_info()

    t = torch.tensor([dt, st["finished"], st["sims"], st["evals"], st["ply_games"]], dtype=torch.float64,
                     device=f"cuda:{local_rank}")
    if world > 1:
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        dt = float(tmax[0])
    finished, sims, evals, ply_games = (float(x) for x in t[1:])
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    hw = n * n
    flop_eval = 2.0 * (3 * 128 * hw + 3 * hw * (128 * 32 + 9 * 32 + 32 * 32 + 32 * 128) + 128 * hw * 512 + 512 * 512 + 512 + 512 * hw)
    flop_fc0 = 2.0 * 128 * hw * 512
    complete = args.max_plies == 0
    mean_plies = ply_games / max(finished, 1.0) if complete else float(plies.mean())
    games_per_s = finished / dt if complete else (ply_games / max(mean_plies, 1.0)) / dt
    # dominant kernel = fc0 GEMM (68 % of the net's MACs): algorithmic flops / HIP-event time (rank 0)
    fc0_s = st["ms_fc0"] * 1e-3
    fc0_tflops = st["fc0_rows"] * flop_fc0 / fc0_s / 1e12 if fc0_s > 0 else 0.0
    net_s = (st["ms_trunk"] + st["ms_fc0"] + st["ms_tail"]) * 1e-3
    round_s = st["ms_tree"] * 1e-3
    out = {
        "metric": "self-play games/sec (15x15, 800 sims/move); MCTS nodes/sec",
        "value": games_per_s, "unit": "games/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / max(args.steps, 1), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f16 (split hi+lo MFMA operands; fc0 correction terms in block-scaled fp6), fp32 accumulate" if args.net_mode == "f16x3" else "f32",
        "data": "synthetic (games from the empty board, random-init net seed 0)",
        "config": {"workload": f"{games} concurrent {n}x{n} games per GPU, {args.sims} sims/move, K={k}, two trees per game"
                               + ("" if complete else f", first {args.max_plies} plies only (games/s extrapolated)"),
                   "games_per_gpu": games, "board": n, "sims_per_move": args.sims, "batch_k": k,
                   "parallelism": f"games sharded x{world}, no hot-path collective" + (", RCCL replay gather" if args.gather else "")},
        "mcts_sims_per_s": sims / dt, "nn_evals_per_s": evals / dt, "plies_per_s": ply_games / dt,
        "mean_plies_per_game": mean_plies, "games_finished": finished,
        "roofline": {"bound": "mfma", "kernel": "k_fc0_mx (fc0)", "achieved": fc0_tflops, "peak": F16_DENSE_PEAK_TFLOPS,
                     "unit": "TFLOP/s", "frac": fc0_tflops / F16_DENSE_PEAK_TFLOPS,
                     "traffic": FC0_HBM_BYTES_PER_ROW.get(n, 0) * st["fc0_rows"] / max(st["fc0_launches"], 1.0) or None,
                     "traffic_unit": "HBM bytes per launch (rows per launch x per-row bytes of the committed PMC pass: "
                                     "profiles/README.md; FETCH_SIZE x2 + WRITE_SIZE)",
                     "algorithmic_bytes_per_launch": (128 * hw * 3 + 2048) * st["fc0_rows"] / max(st["fc0_launches"], 1.0) + 128 * hw * 512 * 3,
                     "mfma_mix_bound": {"value": 1078.0, "unit": "TFLOP/s", "frac_of_bound": fc0_tflops / 1078.0,
                                        "note": "the kernel's MFMA mix alone (operands in registers, random data, one wave per SIMD, every CU): "
                                                "tools/probe/shape_probe mode 2 = 3234 TFLOP/s over the three product terms = 1078 algorithmic"},
                     "note": "algorithmic flops (2*128*HW*512 per eval); per K=64 the kernel issues 4 f16 + 2 block-scaled fp6 "
                             "MFMAs (split operands) = 1.5x the pipe time of a plain-f16 product, so frac <= 0.67 by construction"},
        "roofline_trunk": {"bound": "mfma", "kernel": "k_trunk", "achieved": evals * (flop_eval - flop_fc0 - 2.0 * (512 * 512 + 512 + 512 * hw)) / (st["ms_trunk"] * 1e-3) / 1e12 if st["ms_trunk"] > 0 else 0.0,
                           "peak": F16_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s",
                           "note": "conv_in + 3 bottleneck blocks, 3 f16 MFMAs per product; not MFMA-bound: VALU-issue / LDS / barrier-bound "
                                   "(DESIGN.md 3.2: MFMA busy 36 %, phase model at 82 %)"},
        "roofline_net": {"bound": "mfma", "achieved": evals * flop_eval / net_s / 1e12 if net_s > 0 else 0.0,
                         "peak": F16_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": (evals * flop_eval / net_s / 1e12 / F16_DENSE_PEAK_TFLOPS) if net_s > 0 else 0.0},
        "roofline_tree": {"bound": "hbm", "kernel": "k_round+k_scan+k_scatter", "achieved": st["tree_bytes"] / round_s / 1e9 if round_s > 0 else 0.0,
                          "peak": HBM_PEAK_GBS, "unit": "GB/s",
                          "frac": (st["tree_bytes"] / round_s / 1e9 / HBM_PEAK_GBS) if round_s > 0 else 0.0, "traffic": None},
        "rank0_kernel_ms": {kk: st[kk] for kk in ("ms_round", "ms_tree", "ms_trunk", "ms_fc0", "ms_tail", "ms_ply")},
        "game_length_percentiles": {str(q): float(np.percentile(plies, q)) for q in (0, 10, 25, 50, 75, 90, 99, 100)},
        "arena": {"max_nodes": max_nodes, "max_tables": max_tables, "peak_nodes": st["peak_nodes"], "peak_tables": st["peak_tables"]},
    }
    if complete and world == 1:  # outside the timed region (single-GPU runs only: the other ranks have left by now): the
        # episode-end replay post-processing row on the device
        rec = sp.replay_record_bytes()
        n_rec = 6 * int(plies.sum())
        buf = torch.empty(max(n_rec, 1) * rec, dtype=torch.uint8, device=f"cuda:{local_rank}")
        sp.replay_augment_into(buf.data_ptr(), n_rec)  # warm-up
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        got = sp.replay_augment_into(buf.data_ptr(), n_rec)
        dt_pp = time.perf_counter() - t1
        nw = (hw + 63) // 64
        alg = (n_rec // 6) * (16 * nw + 1 + 4 * hw + 4) + n_rec * rec  # transitions read once + records written
        out["replay_postprocess"] = {"records": got, "ms": 1e3 * dt_pp, "bound": "hbm", "achieved": alg / dt_pp / 1e9, "peak": HBM_PEAK_GBS,
                                     "unit": "GB/s", "frac": alg / dt_pp / 1e9 / HBM_PEAK_GBS,
                                     "note": "z back-fill + 5 augmentations per transition (trainer.rs:207-324), rank 0, host-timed call"}
        if args.train_steps > 0:  # also outside the timed region: the training phase on the same records
            try:
                from omok_ai_amd import train as T
                ph = T.TrainPhase(n, oa.weights.init_random(n, seed=0), f"cuda:{local_rank}")
                ph.run(buf, update_count=2, batch_size=128, seed=0)  # warm-up (MIOpen / rocBLAS plans)
                torch.cuda.synchronize()
                t2 = time.perf_counter()
                v_l, p_l, l_ = ph.run(buf, update_count=args.train_steps, batch_size=128, seed=1)
                torch.cuda.synchronize()
                dt_tr = time.perf_counter() - t2
                out["train_phase"] = {"steps": args.train_steps, "batch": 128, "steps_per_s": args.train_steps / dt_tr, "loss": l_,
                                      "note": "AgentModel::train (Adadelta lr 0.01) via torch autograd on the augmented replay records, rank 0"
                                              " (multi-GPU: gradients averaged by one RCCL all-reduce per step, tests/test_sharding_gloo.py)"}
            except Exception as ex:  # an extra line of the report must never cost the bench line itself
                out["train_phase"] = {"error": repr(ex)}
        del buf
    if args.cpu_seconds > 0 and world == 1:
        try:
            out["cpu_baseline"] = cpu_baseline(args, max(mean_plies, 1.0))
        except Exception as ex:  # (the GPU measurement above stands on its own)
            out["cpu_baseline"] = {"error": repr(ex)}
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
