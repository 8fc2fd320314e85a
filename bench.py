#!/usr/bin/env python3
"""bench.py — self-play throughput of the MI355X engine on BASELINE.json's metric.

A "step" is ONE self-play episode: `games` concurrent 15x15 games per GPU, two agents (trees) per game, `sims` PUCT
simulations per move in rounds of K with one batched net forward per round, random-init net (seed 0), played until every
game has ended (src/trainer.rs:95-205).  Every step runs on its own RNG stream (one reset = one trainer iteration).
`value` = completed games / second over the timed steps, whole job (all ranks).  Warm-up steps are episodes cut after
--warmup-plies plies (untimed: they warm clocks, code objects and allocations; the timed steps are whole episodes).

Default workload = BASELINE.json configs[1]: 4096 concurrent 15x15 games, 800 sims/move, K = 16.
Multi-GPU (configs[3]): one process per GPU, games sharded by global id (game_offset = rank * games), no collective on the
hot path; --gather adds the RCCL all-gather-v of (s, pi, z) replay tuples at episode end (configs[4]).  `python bench.py
--gpus N` starts its N ranks by itself (torch.distributed.run as a child process, before the parent touches the GPU);
started under torchrun it uses the environment it finds.

Output: rank 0 prints the result line as soon as the timed region ends (flush), and -- if the extra legs ran -- the same
line again enriched with `cpu_baseline`, `precision`, `replay_postprocess`, `train_phase`.  The LAST line is the complete
one; the first one exists so that a run killed at its time limit still leaves a measured line.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

T_PROCESS_START = time.perf_counter()
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

F16_DENSE_PEAK_TFLOPS = 2500.0   # MI355X_MICROARCH.md: bf16/f16 MFMA dense
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E spec
BOOST_SCLK_MHZ = 2400.0          # the shader clock the dense peak is quoted at (and what rocm-smi shows on an idle, awake part)
# HBM bytes per net row from the committed rocprofv3 --pmc passes (profiles/README.md): (FETCH_SIZE x 2 [gfx950 128-B
# request correction] + WRITE_SIZE) KiB / rows of the profiled launch.  {board: {kernel: bytes per row}}
PMC_HBM_BYTES_PER_ROW = {15: {"k_trunk": None, "k_fc0_mx": (3.873e6 * 1024 * 2 - 65536 * 28800.0 + 1.347e5 * 1024) / 65536}}
try:  # refreshed by tools/collect_profiles.sh -> profiles/pmc_bytes.json (per-row / per-sim HBM bytes with the guide's corrections)
    with open(os.path.join(ROOT, "profiles", "pmc_bytes.json")) as _f:
        PMC_FILE = json.load(_f)
except Exception:
    PMC_FILE = {}


# ------------------------------------------------------------------------------------------------------------------
# The result line.  The driver parses ONE line of at most a few KB: numbers and short identifiers only.  What every field means is in
# profiles/bench_line_notes.md (keyed by field name); the verbose record of the same run (every probe figure, every leg, the prose) goes to the sidecar file
# gpurun_out/bench_full.json (OMOK_BENCH_FULL overrides the path).  (Round 5's final line had grown to 24 KB and fell out of the driver's record.)
# ------------------------------------------------------------------------------------------------------------------
LINE_LIMIT = 8000


def _r(x, sig=5):
    """numbers of the line carry `sig` significant digits"""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    if isinstance(x, float):
        return float(f"{x:.{sig}g}") if x == x and abs(x) != float("inf") else None
    if isinstance(x, dict):
        return {k: _r(v, sig) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_r(v, sig) for v in x]
    try:
        return _r(float(x), sig)
    except Exception:
        return str(x)[:80]


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and d.get(k) is not None}


ROOF_KEYS = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "executed_flops", "frac_executed", "executed_over_algorithmic", "avg_launch_ms", "rows_per_launch",
             "needed_bytes_per_row", "traffic_over_needed", "mfma_busy_pmc", "valu_busy_pmc", "share_of_kernel_time", "frac_at_sustained_clock",
             "algorithmic_bytes_per_sim", "hbm_bytes_per_sim_pmc", "on_chip_fraction")


def compact_line(full):
    """the driver's line from the verbose record: required keys first, then the measured extras as numbers"""
    line = _pick(full, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling"))
    line["vs_baseline"] = full.get("vs_baseline")
    line.update(_pick(full, ("dtype", "data", "config", "mcts_sims_per_s", "nn_evals_per_s", "plies_per_s", "mean_plies_per_game", "games_finished")))
    for key in ("roofline", "roofline_fc0", "roofline_net", "roofline_tree"):
        if isinstance(full.get(key), dict):
            line[key] = _pick(full[key], ROOF_KEYS)
    if isinstance(full.get("roofline"), dict) and "traffic" not in line["roofline"]:
        line["roofline"]["traffic"] = None
    cb = full.get("cpu_baseline")
    line["cpu_baseline"] = None  # (the contract's key: null on multi-GPU runs and when the leg was switched off; the early line drops nulls)
    if isinstance(cb, dict):
        line["cpu_baseline"] = _pick(cb, ("value", "unit", "cores", "kind", "sample", "host_threads", "host_cpu_quota", "cpu_model", "sims_per_s", "nn_evals_per_s", "net_tflops",
                                          "tree_threads", "net_forward", "one_thread_value", "c1_games_per_s", "error", "skipped"))
    fmt = full.get("fc0_format")
    if isinstance(fmt, dict):
        line["fc0_format"] = {"in_use": fmt.get("in_use_short"), "probe_verdict": (fmt.get("probe_verdict") or {}).get("code")}
    pr = full.get("precision")
    if isinstance(pr, dict):
        cp = _pick(pr, ("within_contract", "error"))
        vs = pr.get("vs_oracle")
        if isinstance(vs, dict):
            num = ("fc0_format", "rows_per_round", "difference_path_rounds", "rows_compared", "rows", "max_dp", "max_dv", "max_dlogit", "max_dvpre", "logit_abs_max")
            cv = _pick(vs, ("error",))
            if isinstance(vs.get("difference_path"), dict):
                cv["difference_path"] = _pick(vs["difference_path"], num)
            if isinstance(vs.get("plain_rows"), dict):
                cv["plain_rows"] = _pick(vs["plain_rows"].get("headline_mode") or {}, num)
            if "north_star_logits_1e-3" in vs:
                cv["north_star_logits_1e-3"] = vs["north_star_logits_1e-3"]
            cp["vs_oracle"] = cv
        for key in ("search_rounds", "search_rounds_timed_size"):
            if isinstance(pr.get(key), dict):
                cp[key] = _pick(pr[key], ("rows", "max_dp", "max_dv", "max_dlogit", "max_dvpre", "within_contract", "children2_launches"))
        line["precision"] = cp
    line.update(_pick(full, ("value_f16_format", "value_fp6_format")))
    for key, keys in (("window_first_plies", ("plies", "ms_per_full_round", "games_per_s_at_mean_length", "mcts_sims_per_s", "error")),
                      ("slots_mode", ("games", "games_per_s", "error")), ("replay_postprocess", ("records", "ms", "achieved", "frac")),
                      ("train_phase", ("steps", "batch", "steps_per_s")), ("replay_gather", ("records_per_episode", "bytes_per_episode", "seconds_per_episode", "last_counts", "last_ids")),
                      ("clocks", ("samples", "sclk_mhz_busy_median", "power_w_busy_median", "sclk_mhz_median", "sclk_mhz_max")),
                      ("children_kernel_launches", ("k_sib_children2", "k_sib_children")), ("rank0_kernel_ms", ("ms_round", "ms_tree", "ms_trunk", "ms_fc0", "ms_tail", "ms_ply"))):
        if isinstance(full.get(key), dict):
            line[key] = _pick(full[key], keys)
    line.update(_pick(full, ("rank0_timed_region_ms", "seconds_since_process_start")))
    line["notes"] = "profiles/bench_line_notes.md"
    line = _r(line)
    if len(json.dumps(line)) > LINE_LIMIT:  # never let an extra cost the record: drop the optional objects, last first
        for key in ("rank0_kernel_ms", "children_kernel_launches", "train_phase", "replay_postprocess", "slots_mode", "window_first_plies", "roofline_tree", "roofline_net", "roofline_fc0", "clocks"):
            line.pop(key, None)
            if len(json.dumps(line)) <= LINE_LIMIT:
                break
    return line


def emit(full, final):
    """print the driver's line (flush) and leave the verbose record beside it"""
    line = compact_line(full)
    if not final:  # the early safety line: a measured line exists from here on whatever happens to the extra legs; it carries no null placeholders
        line = {k: v for k, v in line.items() if v is not None or k == "vs_baseline"}
    path = os.environ.get("OMOK_BENCH_FULL") or os.path.join(ROOT, "gpurun_out", "bench_full.json")
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            json.dump(full, f, default=lambda o: float(o) if hasattr(o, "__float__") else str(o))
    except OSError:
        pass
    print(json.dumps(line), flush=True)



def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=0)
    ap.add_argument("--warmup-plies", type=int, default=4, help="plies of a warm-up step (0 = whole episodes)")
    ap.add_argument("--board", type=int, default=15)
    ap.add_argument("--games", type=int, default=4096, help="concurrent games per GPU")
    ap.add_argument("--sims", type=int, default=800)
    ap.add_argument("--batch-k", type=int, default=16, help="evaluate_batch_size (src/config.rs:92)")
    ap.add_argument("--max-plies", type=int, default=0, help="debug: stop every episode after this many plies")
    ap.add_argument("--max-nodes", type=int, default=0)
    ap.add_argument("--max-tables", type=int, default=0)
    ap.add_argument("--net-mode", default="f16x3", choices=["f16x3", "fp6", "mixed", "f16", "f32"],
                    help="f16x3: split-operand MFMA, fc0's correction terms in the format omok_net_commit's probe keeps (the headline); fp6 / f16: that format forced "
                         "(f16 meets north_star's 1e-3 on the LOGITS too); f32: the fp32 VALU kernels")
    ap.add_argument("--f16-leg", type=int, default=1, help="extra legs: one more whole episode each with fc0 forced into the f16 / the fp6 operand format -> value_f16_format, value_fp6_format (0 = skip)")
    ap.add_argument("--gather", action="store_true", help="RCCL all-gather-v of replay tuples at episode end")
    ap.add_argument("--cpu-seconds", type=float, default=80.0, help="budget of the cpu_baseline leg (0 = skip)")
    ap.add_argument("--train-steps", type=int, default=20, help="training steps timed after the episode (0 = skip; batch 128)")
    ap.add_argument("--slots-multiple", type=int, default=3, help="extra leg (outside the timed region): slots mode, this many x games played on the "
                    "engine's game slots with finished slots restarted (omok_selfplay_run_slots); 0 = skip")
    ap.add_argument("--profile-every", type=int, default=8, help="HIP-event timing of the kernel categories on 1 search round in N (sums scaled); "
                    "an event record idles the queue ~5 us, 6 category boundaries per round")
    ap.add_argument("--precision-rows", type=int, default=4096, help="rows of the in-run precision check (0 = skip)")
    ap.add_argument("--budget-seconds", type=float, default=float(os.environ.get("OMOK_BENCH_BUDGET_S", "540")),
                    help="wall-clock budget of the whole process: the extra legs only run while there is room")
    ap.add_argument("--window-plies", type=int, default=20, help="extra leg: time the first N plies of one more episode (SURVEY 8d's fixed-length window; 0 = skip)")
    ap.add_argument("--seed", type=int, default=0)
    return ap.parse_args(argv)


def elapsed():
    return time.perf_counter() - T_PROCESS_START


# ------------------------------------------------------------------------------------------------------------------
# clock / power samples of the timed region (rank 0): a child process started BEFORE this process touches the GPU runs `rocm-smi` about twice a second and appends
# "unix time, sclk MHz, W" (of the busiest GPU) to a scratch file; the lines between the two barriers of the timed region are summarised into the JSON line.
# Every kernel of the search rounds runs against the package power limit on this part (profiles/r05_power_by_kernel.txt), so the clock the peak is quoted at is not the clock the run gets.
# ------------------------------------------------------------------------------------------------------------------
CLOCK_SAMPLER = r"""
import os, re, subprocess, sys, time
out = open(sys.argv[1], "a", buffering=1)
parent, t_end = int(sys.argv[2]), time.time() + 3600.0
while os.getppid() == parent and time.time() < t_end:  # (never outlives the bench process that started it)
    try:
        txt = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=10).stdout
    except Exception:
        break
    sclk, watt = {}, {}
    for line in txt.splitlines():
        m = re.match(r"GPU\[(\d+)\]\s*:\s*sclk clock level.*\((\d+)Mhz\)", line)
        if m: sclk[m.group(1)] = int(m.group(2))
        m = re.match(r"GPU\[(\d+)\]\s*:.*Power \(W\):\s*([0-9.]+)", line)
        if m: watt[m.group(1)] = float(m.group(2))
    if not sclk or not watt:
        break
    g = max(watt, key=watt.get)
    out.write(f"{time.time():.3f} {sclk.get(g, 0)} {watt[g]:.0f}\n")
    time.sleep(0.25)
"""


def start_clock_sampler():
    import shutil
    import tempfile
    if os.environ.get("OMOK_BENCH_CLOCKS", "1") == "0" or not shutil.which("rocm-smi"):
        return None
    # Under a profiler the GPU is already initialised when this program starts (rocprofv3 --pmc preloads its tool library, which opens the device), and a process that has
    # initialised the GPU must not start a program: no sampler then.
    if "rocprof" in os.environ.get("LD_PRELOAD", "").lower() or any(k.startswith(("ROCP_", "ROCPROF", "ROCTRACER", "HSA_TOOLS_LIB")) for k in os.environ):
        return None
    try:
        path = os.path.join(tempfile.gettempdir(), f"omok_bench_clocks_{os.getpid()}.txt")
        open(path, "w").close()
        return subprocess.Popen([sys.executable, "-c", CLOCK_SAMPLER, path, str(os.getpid())], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL), path
    except Exception:
        return None


def stop_clock_sampler(sampler, t_begin, t_end):
    if not sampler:
        return None
    proc, path = sampler
    try:
        proc.terminate()  # (this exact child)
        proc.wait(timeout=15)
    except Exception:
        pass
    try:
        rows = [[float(x) for x in l.split()] for l in open(path) if len(l.split()) == 3]
        os.remove(path)
    except Exception:
        return None
    rows = [r for r in rows if t_begin <= r[0] <= t_end]
    if len(rows) < 4:
        return None
    import numpy as np
    sclk, watt = np.array([r[1] for r in rows]), np.array([r[2] for r in rows])
    busy = watt >= 0.85 * np.percentile(watt, 90)  # (samples inside well-filled rounds: the thin tail of an episode draws a fraction of the power and runs at the boost clock)
    return {"method": "rocm-smi (--showclocks --showpower) from a child process, ~2 samples per second between the barriers of the timed region",
            "samples": len(rows), "sclk_mhz_median": float(np.median(sclk)), "sclk_mhz_p10": float(np.percentile(sclk, 10)), "sclk_mhz_max": float(sclk.max()),
            "power_w_median": float(np.median(watt)), "power_w_max": float(watt.max()),
            "busy_samples": int(busy.sum()), "sclk_mhz_busy_median": float(np.median(sclk[busy])), "power_w_busy_median": float(np.median(watt[busy])),
            "note": "busy = samples at >= 0.85 x the 90th percentile of the power readings, i.e. inside the well-filled rounds: there the package sits at its power limit and the shader "
                    "clock below the boost clock the peaks are quoted at (profiles/r05_power_by_kernel.txt: each kernel of a round alone runs at the limit too)"}


# ------------------------------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` without a torchrun environment
# ------------------------------------------------------------------------------------------------------------------
def self_launch(args):
    """Start N ranks as a CHILD process (torch.distributed.run) and relay its exit code.  Nothing in this process has
    touched the GPU yet (torch.cuda.device_count() does not initialise it on this image)."""
    import torch
    backend = os.environ.get("OMOK_BENCH_BACKEND", "nccl")
    have = torch.cuda.device_count() if backend == "nccl" else args.gpus  # (gloo rehearsal: ranks may share a GPU)
    if have < args.gpus:
        msg = f"--gpus {args.gpus}: {args.gpus} GPUs needed, {have} visible on this host; nothing was run"
        print(json.dumps({"error": msg, "n_gpus": args.gpus, "gpus_visible": have, "value": None}), flush=True)
        print(msg, file=sys.stderr)
        return 3  # (nothing was measured: not a success)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


# ------------------------------------------------------------------------------------------------------------------
# extra legs (outside the timed region, rank 0 of a 1-GPU run only)
# ------------------------------------------------------------------------------------------------------------------
def cpu_baseline(args, mean_plies, budget_s):
    """the CPU restatement of the reference path on this host's cores: oracle tree code (C) + a BLAS-backed fp32
    forward (torch-CPU on the graph of omok-ai_amd/train.py), kind "port".  Workloads: C1 exactly (one 15x15 game, 100
    sims/move -> 112 with K = 16, played to the end or to the leg's time share) and C2' (256 games, 800 sims/move, first
    plies, >= 64 threads; 16 games on 1 thread), every leg bounded by its share of `budget_s`."""
    import numpy as np
    import torch
    from oracle import oracle as O
    import omok_ai_amd as oa
    from omok_ai_amd import train as T
    n, k = args.board, args.batch_k
    hw = n * n
    cores = os.cpu_count() or 1
    tensors = oa.weights.init_random(n, seed=0)
    net = T.Network(n, tensors, "cpu", allow_cpu=True)

    onet_c = O.Net(n, tensors)

    class MMForward:
        """The same fp32 forward as matrix products (network.rs:51-247): every 1x1 convolution is `rows*HW x Cin @ Cin x Cout` (torch.addmm -> MKL sgemm), the depthwise
        3x3 nine shifted multiply-adds on the NHWC grid, fc0 one [rows, 128 HW] x [128 HW, 512] product -- the shape a CPU BLAS is good at (VERDICT round 5, item 8:
        the conv2d graph of train.py reached 2.9 k rows/s whatever the thread count).  Rows in chunks of `cs` (chosen by the calibration below: small chunks keep the
        115 KB-per-row activations in cache, large ones give MKL more rows per product)."""
        cs = 256

        def __init__(self, row_parallel):
            self.row_parallel = row_parallel  # False: whole chunks of 256 rows one after the other on torch's intra-op threads (the better one on a few cores)
            sh = oa.weights.tensor_shapes(n)
            t = [torch.as_tensor(np.asarray(x, dtype=np.float32).reshape(s_)) for x, s_ in zip(tensors, sh)]
            self.w_in, self.b_in = t[0].reshape(3, 128).contiguous(), t[1]
            self.blocks = []
            for i in range(3):
                w0, b0, dw, pw, b1, w2, b2 = t[2 + 7 * i: 9 + 7 * i]
                self.blocks.append((w0.reshape(128, 32).contiguous(), b0, dw.reshape(9, 32).contiguous(), pw.reshape(32, 32).contiguous(), b1, w2.reshape(32, 128).contiguous(), b2))
            self.fc = t[23:31]

        def chunk(self, x):
            import torch.nn.functional as F
            b = x.shape[0]
            a = F.leaky_relu_(torch.addmm(self.b_in, x.reshape(b * hw, 3), self.w_in), 0.2)
            for w0, b0, dw, pw, b1, w2, b2 in self.blocks:
                h = F.leaky_relu_(torch.addmm(b0, a, w0), 0.2)
                hp = F.pad(h.view(b, n, n, 32), (0, 0, 1, 1, 1, 1))
                d = hp[:, 0:n, 0:n, :] * dw[0]
                for tap in range(1, 9):  # taps in (dy, dx) order like the reference's depthwise (network-utils/src/lib.rs:172-262)
                    d.addcmul_(hp[:, tap // 3: tap // 3 + n, tap % 3: tap % 3 + n, :], dw[tap])
                g = F.leaky_relu_(torch.addmm(b1, d.reshape(b * hw, 32), pw), 0.2)
                a = F.leaky_relu_(torch.addmm(b2, g, w2).add_(a), 0.2)  # add before the activation (network.rs:108-111)
            w_fc0, b_fc0, w_fc1, b_fc1, w_v, b_v, w_p, b_p = self.fc
            h0 = F.leaky_relu_(torch.addmm(b_fc0, a.view(b, hw * 128), w_fc0), 0.2)
            h1 = F.leaky_relu_(torch.addmm(b_fc1, h0, w_fc1), 0.2)
            return torch.softmax(torch.addmm(b_p, h1, w_p), dim=1), torch.tanh(torch.addmm(b_v, h1, w_v))

        def __call__(self, x, threads):
            """rows dealt in chunks of `cs` to `threads` workers, every worker running its matrix products on ONE MKL thread: torch's intra-op pool does not scale the
            trunk's small products (on the GPU box's 2 x 64 cores the same graph with 64 intra-op threads ran 3.4 k rows/s against 10.8 k of the row-parallel C loops)"""
            xt = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).reshape(len(x), hw * 3)
            starts = list(range(0, len(xt), self.cs))
            with torch.no_grad():
                if threads <= 1 or len(starts) == 1 or not self.row_parallel:
                    outs = [self.chunk(xt[i:i + self.cs]) for i in starts]
                else:
                    outs = list(self.pool(threads).map(lambda i: self.chunk(xt[i:i + self.cs]), starts))
            return torch.cat([o[0] for o in outs]).numpy(), torch.cat([o[1] for o in outs]).numpy().reshape(-1)

        pools = {}

        def pool(self, threads):
            from concurrent.futures import ThreadPoolExecutor
            if threads not in self.pools:  # (a worker's OpenMP / MKL team size is its own: set once, when the worker starts)
                self.pools[threads] = ThreadPoolExecutor(max_workers=threads, initializer=lambda: torch.set_num_threads(1))
            return self.pools[threads]

    mm_net, mm_intra = MMForward(True), MMForward(False)


    def forward_torch(x, threads):
        with torch.no_grad():
            p, v = net(torch.from_numpy(np.ascontiguousarray(x)).reshape(-1, n, n, 3))
        return p.numpy().reshape(len(x), hw), v.numpy().reshape(-1)

    def forward_c(x, threads):  # the oracle's own fp32 forward (oracle/net.c: plain loops, OpenMP over blocks of 8 rows)
        return onet_c.forward(np.ascontiguousarray(x, dtype=np.float32).reshape(len(x), -1), threads=threads)

    FORWARDS = {"torch-CPU mm (MKL sgemm, row-parallel)": mm_net, "torch-CPU mm (MKL sgemm, intra-op threads)": mm_intra, "oracle/net.c (OpenMP over rows)": forward_c}
    if os.environ.get("OMOK_BENCH_CONV2D"):  # (train.py's conv2d graph: 2.9 k rows/s on the GPU box's host at any thread count, rounds 5 and 6 -- not worth its calibration time)
        FORWARDS["torch-CPU (conv2d graph)"] = forward_torch

    seen_rows = []  # request rows of the legs' real search rounds (for the oracle-vs-GPU check of the net outputs below)

    def leg(games, sims, threads, seconds, max_plies, engine="torch-CPU mm (MKL sgemm, row-parallel)"):
        torch.set_num_threads(1 if engine.endswith("row-parallel)") else threads)  # (the matrix-product forward runs its own workers, one MKL thread each)
        fwd = FORWARDS[engine]

        def forward(x):
            return fwd(x, threads)
        root_p, _ = forward(O.Environment(n).encode_nn_input(0)[None])
        sp = O.SelfPlay(n, games, cap_nodes=min(16384, 4 * sims + 1024), cap_tables=max(256, sims + 256), seed=args.seed)
        sp.set_threads(min(threads, games))  # the tree loops over the games under OpenMP (the reference: rayon par_iter, parallel_mcts_executor.rs:200-205)
        sp.reset(root_p[0])
        rounds = (sims + k - 1) // k
        t0 = time.perf_counter()
        t_net = 0.0
        n_sims = n_evals = 0
        plies = 0
        out_of_time = False
        while sp.alive_count > 0 and plies < max_plies and not out_of_time:
            for rnd in range(rounds):
                inp = sp.round_generate(rnd, k, 0.25, 0.03)
                n_sims += k * sp.alive_count
                if len(inp):
                    if games > 1 and sum(len(r) for r in seen_rows) < 16384:
                        seen_rows.append(np.array(inp[:: max(1, len(inp) // 32)], dtype=np.float32))  # (a spread of every round's rows)
                    t1 = time.perf_counter()
                    p, v = forward(inp)
                    t_net += time.perf_counter() - t1
                    n_evals += len(inp)
                    sp.round_scatter(p, v)
                if time.perf_counter() - t0 > seconds:
                    out_of_time = True
                    break
            if out_of_time:
                break
            sp.sample(1.0, 30)
            m = sp.mirror_generate()
            t1 = time.perf_counter()
            p, _ = forward(m)
            t_net += time.perf_counter() - t1
            n_evals += len(m)
            sp.advance(p)
            plies += 1
        dt = time.perf_counter() - t0
        return {"games": games, "sims_per_move": rounds * k, "threads": threads, "net_forward": engine, "seconds": dt, "plies_completed": plies,
                "sims": n_sims, "sims_per_s": n_sims / dt, "nn_evals": n_evals, "nn_evals_per_s": n_evals / dt, "net_seconds": t_net,
                "tree_seconds": dt - t_net, "finished": sp.alive_count == 0}

    # C2' (SURVEY 8d): 256 games -> 4096-row forwards per round, the batch shape at which a BLAS-backed CPU forward is efficient, on at least 64 of the host's threads
    # (the fastest of {64, 128, all}; a host with fewer threads uses all of them).  Three forwards compete at every thread count: the matrix-product formulation on MKL
    # (VERDICT round 5, item 8), train.py's conv2d graph (2.9 k rows/s on the GPU box's 2 x 64-core host whatever the thread count: it does not use the machine) and the
    # oracle's own row-parallel C loops.  The tree part is the oracle's C code with the loops over the games under OpenMP on the same threads (the reference: a rayon pool).
    # The one-thread legs keep 16 games.  Budget: 15 % calibration, 8 % per C1 leg, 50 % the C2' leg, 12 % the one-thread C2' leg.
    # what the process may use of the host: the cgroup's CPU quota (the GPU pool's boxes grant 16 CPUs of a 2 x 64-core host: cpu.max "1600000 100000", measured round 6 --
    # 64 workers then run at a quarter of their speed each) and the affinity mask
    quota = None
    try:
        q_, per_ = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        quota = None if q_ == "max" else max(1, int(round(float(q_) / float(per_))))
    except (OSError, ValueError):
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        usable = cores
    if quota and quota < usable:
        cand = sorted({t for t in (quota, 2 * quota, 4 * quota) if t <= usable})
    else:
        cand = sorted({t for t in (64, 128, usable) if t <= usable}) or [usable]
    try:  # glibc: keep the forward's multi-megabyte temporaries in the arenas (every mmap / munmap / page fault of 16+ workers in one process goes through one lock)
        import ctypes
        libc_ = ctypes.CDLL("libc.so.6")
        libc_.mallopt(-3, 1 << 30)  # M_MMAP_THRESHOLD
        libc_.mallopt(-1, 1 << 30)  # M_TRIM_THRESHOLD
    except Exception:
        pass
    g2, g1t = 256, 16
    flop_eval = 2.0 * (3 * 128 * hw + 3 * hw * (128 * 32 + 9 * 32 + 32 * 32 + 32 * 128) + 128 * hw * 512 + 512 * 512 + 512 + 512 * hw)
    # does MKL use the machine?  fc0's product alone ([1024, 128 HW] x [128 HW, 512]) at 1 thread and at every candidate count, ~1 s in all: TFLOP/s must scale
    mm_cal = {}
    xa, wb = torch.randn(1024, 128 * hw), torch.randn(128 * hw, 512)
    for t in [1] + cand:
        torch.set_num_threads(t)
        t_all, best_call, calls = time.perf_counter(), 1e9, 0
        while calls < 3 or time.perf_counter() - t_all < (1.2 if t > 1 else 0.4):  # (the best single call: a new thread team's first ~0.5 s of products run at a fraction of its rate)
            t1 = time.perf_counter()
            torch.mm(xa, wb)
            best_call = min(best_call, time.perf_counter() - t1)
            calls += 1
        mm_cal[str(t)] = 2.0 * 1024 * 128 * hw * 512 / best_call / 1e12
    del xa, wb
    # chunk size of the matrix-product forward (rows per task of a worker): two 4096-row forwards per candidate on the smallest candidate thread count
    torch.set_num_threads(1)
    xv = (np.random.RandomState(1).rand(4096, 3 * hw) < 0.2).astype(np.float32)
    mm_chunks = {}
    for cs in (64, 128, 256):
        if 4096 // cs < cand[0]:
            continue  # (fewer tasks than workers)
        mm_net.cs = cs
        mm_net(xv, cand[0])
        t1 = time.perf_counter()
        mm_net(xv, cand[0])
        mm_chunks[cs] = 4096 / (time.perf_counter() - t1)
    mm_net.cs = max(mm_chunks, key=mm_chunks.get) if mm_chunks else 64
    # the matrix-product forward against the oracle's own (the checker of the -m gpu tests) on 64 deterministic rows
    xv = (np.random.RandomState(0).rand(64, 3 * hw) < 0.2).astype(np.float32)
    pm, vm = mm_net(xv, min(8, cores))
    pc, vc = forward_c(xv, min(8, cores))
    mm_check = {"rows": 64, "max_dp": float(np.abs(pm - pc).max()), "max_dv": float(np.abs(vm - vc).max())}
    calib_legs = {(e, t): leg(g2, args.sims, t, max(0.6, 0.15 * budget_s / (len(FORWARDS) * len(cand))), 1, e) for e in FORWARDS for t in cand}
    calib = {kk: v["sims_per_s"] for kk, v in calib_legs.items()}
    best_e, best_t = max(calib, key=calib.get)
    share = 0.50 * budget_s
    legs = {
        "c1_best_threads": leg(1, 100, best_t, 0.08 * budget_s, 10 ** 6, best_e),
        "c1_one_thread": leg(1, 100, 1, 0.08 * budget_s, 10 ** 6, best_e),
        "c2p_best_threads": leg(g2, args.sims, best_t, share, 4, best_e),
        "c2p_one_thread": leg(g1t, args.sims, 1, 0.12 * budget_s, 2, best_e),
    }
    torch.set_num_threads(cores)
    best = legs["c2p_best_threads"]
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    rounds_up = (args.sims + k - 1) // k * k
    # The oracle's own net forward (oracle/net.c, fp32, the checker of the -m gpu parity tests) on 256 request rows of the rounds above: timed as a
    # sample of the CPU path's net part, and its outputs handed back so that main() can state the GPU's |dp|, |dv| and |dlogit| against the ORACLE
    net_check = None
    if seen_rows:
        allr = np.concatenate(seen_rows)
        xr = np.ascontiguousarray(allr[:: max(1, len(allr) // 256)][:256])
        onet = onet_c
        t1 = time.perf_counter()
        po, vo, lgo, vpo = onet.forward_logits(xr, threads=best_t)
        dt_o = time.perf_counter() - t1
        net_check = {"x": xr, "p": po, "v": vo, "logits": lgo, "vpre": vpo, "seconds": dt_o, "threads": best_t}
    return {"_net_check": net_check,
            "oracle_net_forward": ({"rows": int(len(net_check["x"])), "seconds": net_check["seconds"], "rows_per_s": len(net_check["x"]) / net_check["seconds"],
                                    "threads": best_t, "what": "oracle/net.c fp32 forward (plain loops, OpenMP over blocks of 8 rows) on request rows of the legs' search rounds"}
                                   if net_check else None),
            "value": best["sims_per_s"] / (rounds_up * mean_plies), "unit": "games/s", "cores": best_t, "host_threads": cores, "kind": "port",
            "net_forward": best_e, "tree_threads": min(best_t, g2), "host_cpu_quota": quota, "host_affinity_cpus": usable,
            "net_tflops": best["nn_evals"] * flop_eval / max(best["net_seconds"], 1e-9) / 1e12,  # the forward alone: evaluations x 2*MAC flops / time inside the forward
            "net_share_of_time": best["net_seconds"] / max(best["seconds"], 1e-9), "nn_evals_per_s": best["nn_evals_per_s"],
            "mm_calibration_tflops_by_threads": mm_cal, "mm_forward_rows_per_s_by_chunk": mm_chunks, "mm_forward_vs_oracle": mm_check,
            "thread_calibration_sims_per_s": {f"{e} @ {t} threads": v for (e, t), v in calib.items()},
            "sample": f"C2': {g2} games x {rounds_up} sims x {best['plies_completed']} plies ({best['seconds']:.0f} s), {best_t} threads on {quota if quota else usable} granted CPUs of {cores}",
            "sample_long": f"C2' = {g2} games x {rounds_up} sims/move ({g2 * k}-row forwards) x up to 4 plies (bounded to {share:.0f} s; {best['plies_completed']} plies completed) on {best_t} of {cores} "
                           f"threads (fastest of {cand} x the forwards of thread_calibration_sims_per_s): oracle C tree code (games under OpenMP) + fp32 forward by {best_e}; {best['sims_per_s']:.0f} sims/s, converted with {mean_plies:.1f} plies/game "
                           f"from the GPU run.  Also C1 (1 game, 100->112 sims/move, whole game or {0.08 * budget_s:.0f} s) and, on 1 thread, C1 and {g1t} games of C2': see legs",
            "cpu_model": model, "sims_per_s": best["sims_per_s"],
            "one_thread_value": legs["c2p_one_thread"]["sims_per_s"] / (rounds_up * mean_plies),
            "c1_games_per_s": {kk: (1.0 / v["seconds"] if v["finished"] else v["sims_per_s"] / (112 * mean_plies)) for kk, v in legs.items() if kk.startswith("c1")},
            "legs": legs}


def precision_check(args, rows, device):
    """Precision evidence inside the run: `rows` request rows of real self-play rounds (positions 6 and 14 plies into games
    of this net), evaluated by the product path (f16 split operands, fp6 corrections) and by the OMOK_NET_F32 kernels on the
    same GPU.  Reports the outputs the reference API returns (p after softmax, v after tanh: the 1e-3 contract) and the
    quantities in front of the last ops (logits, pre-tanh value)."""
    import numpy as np
    import omok_ai_amd as oa
    from omok_ai_amd import binding as B
    n, k = args.board, args.batch_k
    games = max(8, rows // (2 * k))
    eng = oa.Engine(board_size=n, games=games, max_nodes=2048, max_tables=512, max_batch_k=k, device=device, seed=args.seed + 1)
    eng.load_random_weights(0)
    sp = oa.SelfPlay(eng)
    sp.reset()
    xs = []
    for stop in (6, 14):
        sp.run(64, k, 0.25, 0.03, 1.0, 30, stop - sp.ply)
        if sp.alive_count == 0:
            break
        sp.round_generate(0, k)
        xs.append(sp.round_inputs())
        sp.round_eval()
        sp.round_scatter()
        for rnd in range(1, 4):
            sp.round_generate(rnd, k)
            sp.round_eval()
            sp.round_scatter()
        sp.sample_actions(1.0, 30)
        sp.advance()
    x = np.concatenate(xs)[:rows]
    eng.close()
    out = oa.precision.measure(oa.weights.init_random(n, seed=0), n, x, device=device, batch_k=k)
    # the search rounds' own path through trunk and fc0 (N = 15: sibling base + window differences), on the rows of real rounds
    out["search_rounds"] = oa.precision.measure_search_rounds(oa.weights.init_random(n, seed=0), n, games=512, batch_k=k,
                                                              rounds=4, plies=2, device=device, seed=args.seed + 2)  # (8192-row rounds)
    out["within_contract"] = bool(out["within_contract"] and out["search_rounds"]["within_contract"])
    if args.games * k > 8192:  # one ply of rounds at the size that is TIMED (configs[1]: 65536 rows per round: multi-tile window bins, the K-split set)
        out["search_rounds_timed_size"] = oa.precision.measure_search_rounds(oa.weights.init_random(n, seed=0), n, games=args.games, batch_k=k,
                                                                             rounds=3, plies=1, device=device, seed=args.seed + 3)
        out["within_contract"] = bool(out["within_contract"] and out["search_rounds_timed_size"]["within_contract"])
    out["reference"] = "OMOK_NET_F32 kernels on the same GPU (fp32 VALU, k-ascending sums)"
    out["contract"] = "1e-3 on the outputs of AgentModel::evaluate_pv (p after softmax, v after tanh)"
    return out


# ------------------------------------------------------------------------------------------------------------------
def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "0") or 0)
    if world == 0 and args.gpus > 1:
        sys.exit(self_launch(args))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = max(world, 1)
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: using the launcher's world size", file=sys.stderr)

    sampler = start_clock_sampler() if rank == 0 else None  # (a child process: started before anything here initialises the GPU)

    import numpy as np
    import torch
    import torch.distributed as dist

    backend = os.environ.get("OMOK_BENCH_BACKEND", "nccl")
    mock = "OMOK_BENCH_ENGINE" in os.environ
    use_cuda = not mock  # the engine's device; with the gloo backend (rehearsals on a 1-GPU box) the collectives run on CPU tensors
    ndev = torch.cuda.device_count() if use_cuda else 0
    gpu = local_rank % ndev if ndev else 0  # (a gloo rehearsal may put several ranks on one GPU; RCCL runs never do)
    device = f"cuda:{gpu}" if backend == "nccl" else "cpu"
    if use_cuda:
        torch.cuda.set_device(gpu)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", gpu))
        else:
            dist.init_process_group(backend)

    import importlib
    oa = importlib.import_module(os.environ.get("OMOK_BENCH_ENGINE", "omok_ai_amd"))  # (tests substitute a GPU-free stand-in)
    B = oa.binding

    n, games, k = args.board, args.games, args.batch_k
    max_nodes = args.max_nodes or min(16384, 4 * args.sims + 1024)
    max_tables = args.max_tables or max(256, max_nodes // 4)
    eng = oa.Engine(board_size=n, games=games, max_nodes=max_nodes, max_tables=max_tables, max_batch_k=k,
                    device=gpu, net_mode={"f16x3": B.NET_F16X3, "fp6": B.NET_F16X3_FP6, "mixed": B.NET_F16X3_MIXED, "f16": B.NET_F16X3_F16, "f32": B.NET_F32}[args.net_mode],
                    seed=args.seed, game_offset=oa.dist.game_offset(rank, games))
    eng.load_random_weights(0)
    sp = oa.SelfPlay(eng)
    eng.set_profiling(max(1, args.profile_every))

    def barrier():
        if use_cuda:
            torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        if use_cuda:
            torch.cuda.synchronize()

    rec = sp.replay_record_bytes()
    gather_buf = torch.empty(games * n * n * rec, dtype=torch.uint8, device=f"cuda:{gpu}" if use_cuda else "cpu") if args.gather else None
    gathered = {"records": 0, "bytes": 0, "seconds": 0.0}

    def episode(max_plies):
        sp.reset()
        st = sp.run(args.sims, k, 0.25, 0.03, 1.0, 30, max_plies)
        if args.gather:
            t1 = time.perf_counter()
            cnt = sp.pack_tensor(gather_buf, rec) if hasattr(sp, "pack_tensor") else sp.replay_pack_into(gather_buf.data_ptr(), gather_buf.numel() // rec)
            live = gather_buf[: cnt * rec].view(cnt, rec)
            allrec, counts = oa.dist.gather_replay(live if backend == "nccl" or not use_cuda else live.cpu())  # counts exchange + exact-size all-gather-v
            if use_cuda:
                torch.cuda.synchronize()
            gathered["last_counts"] = counts
            gathered["last_ids"] = allrec[:, :8].contiguous().view(torch.int64).reshape(-1).tolist() if os.environ.get("OMOK_MOCK_DIR") else None
            gathered["records"] += int(allrec.shape[0])
            gathered["bytes"] += int(allrec.numel())
            gathered["seconds"] += time.perf_counter() - t1
        return st

    for _ in range(args.warmup):
        episode(args.warmup_plies if args.max_plies == 0 else min(args.warmup_plies or args.max_plies, args.max_plies))
    eng.reset_stats()
    for key in ("records", "bytes", "seconds"):
        gathered[key] = 0
    barrier()
    t0, wall0 = time.perf_counter(), time.time()
    for _ in range(args.steps):
        episode(args.max_plies)
    barrier()
    dt = time.perf_counter() - t0
    clocks = stop_clock_sampler(sampler, wall0, time.time())
    st = eng.stats()
    alive, status, plies = sp.game_info()

    t = torch.tensor([dt, st["finished"], st["sims"], st["evals"], st["ply_games"]], dtype=torch.float64, device=device)
    if world > 1:
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        dt = float(tmax[0])
    finished, sims, evals, ply_games = (float(x) for x in t[1:])
    if rank != 0:
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    hw = n * n
    flop_eval = 2.0 * (3 * 128 * hw + 3 * hw * (128 * 32 + 9 * 32 + 32 * 32 + 32 * 128) + 128 * hw * 512 + 512 * 512 + 512 + 512 * hw)
    flop_fc0 = 2.0 * 128 * hw * 512
    flop_tail = 2.0 * (512 * 512 + 512 + 512 * hw)
    flop_trunk = flop_eval - flop_fc0 - flop_tail
    complete = args.max_plies == 0
    mean_plies = ply_games / max(finished, 1.0) if complete else float(plies.mean())
    games_per_s = finished / dt if complete else (ply_games / max(mean_plies, 1.0)) / dt
    rows = st["fc0_rows"]                       # net rows evaluated on rank 0 in the timed region
    launches = max(st["fc0_launches"], 1.0)     # one trunk + one fc0 launch per forward
    k_ms = {"k_trunk": st["ms_trunk"], "k_fc0_mx": st["ms_fc0"], "tail (fc1, heads, softmax)": st["ms_tail"],
            "tree (k_round, k_scan, k_scatter)": st["ms_tree"], "ply (sample, mirror, advance)": st["ms_ply"]}
    dominant = max(("k_trunk", "k_fc0_mx"), key=lambda kk: k_ms[kk])
    pmc = PMC_FILE.get(str(n), {})

    # ---- executed matrix work: MFMA instructions counted from the kernels' structure (tools/isa_hist.py --mfma confirms the per-pass counts in the
    #      code object) x the engine's device-side work counters (OMOK_STAT_WORK_*), beside the algorithmic credit.  Every MFMA of this library -- v_mfma_f32_32x32x16_f16
    #      (32768 flop) and the block-scaled fp6 v_mfma_scale_f32_32x32x64_f8f6f4 (131072 flop) -- occupies a SIMD's matrix pipe for 32 cycles (MI355X_MICROARCH.md), so
    #      frac_executed = MFMAs x 32 cycles / (time x 1024 SIMDs x 2.4 GHz): the pipe-busy fraction at the boost clock, comparable with mfma_busy_pmc.
    tiles_row = (hw + 31) // 32                      # 32-pixel MFMA tiles of a full trunk row (k_trunk: one wave per tile)
    M_TRUNK_TILE, M_CHILD2, M_CHILD1 = 170, 200, 340  # MFMAs: k_trunk per tile (conv_in 8 + 3 blocks x (24 + 6 + 24)); k_sib_children2 per child; k_sib_children per child (2 tiles)
    M_MX_STEP, M_X3_HALF = 4 * 96, 4 * 96             # per 512 x 128 fc0 tile (4 waves): k_fc0_mx super-step K = 64 (64 f16 + 32 fp6 per wave); k_fc0_x3 half-step K = 32 (96 f16 per wave)
    F_MX_STEP, F_X3_HALF = 4 * (64 * 32768 + 32 * 131072), 4 * 96 * 32768
    M_FC1_TILE = 32 * 24 * 8                         # k_gemm_t<16>: 32 k-steps x 24 MFMAs per wave x 8 waves per 128-row tile
    heads_mt = 8 if n == 15 else 4                   # k_gemm_t<MT>: output features / 32 (226 -> 256 at N = 15, 82 -> 128 at N = 9)
    M_HEADS_TILE = 32 * (heads_mt * 4 * 3 // 8) * 8  # 32 k-steps x (MT x 4 accumulator tiles x 3 terms / 8 waves) x 8 waves
    wk = {kk: float(st.get(kk, 0.0) or 0.0) for kk in ("work_diff_runs", "work_diff_singles", "work_diff_children", "work_copy_runs", "work_copy_singles", "work_copy_children",
                                                       "work_diff_full_runs", "work_win_pixels", "work_win_tiles", "work_full_tiles")}
    have_work = "work_diff_children" in st
    fcode_all = int(st.get("fc0_format", 0))
    plain_rows = max(0.0, rows - wk["work_diff_children"] - wk["work_diff_singles"] - wk["work_copy_children"] - wk["work_copy_singles"])
    full_trunk_rows = wk["work_diff_full_runs"] + wk["work_diff_singles"] + wk["work_copy_runs"] + wk["work_copy_singles"] + plain_rows
    mfma_trunk = M_CHILD2 * wk["work_diff_children"] + M_CHILD1 * wk["work_copy_children"] + M_TRUNK_TILE * tiles_row * full_trunk_rows
    dense_tiles = (rows - wk["work_diff_children"] - wk["work_diff_singles"]) / 128.0   # rows of the copy path and of plain forwards go through the dense fc0
    if fcode_all == 0:    # fp6 rows: k_fc0_mx everywhere
        m_full, f_full = 2 * hw * M_MX_STEP, 2 * hw * F_MX_STEP
    else:                 # f16 full rows (f16 and mixed formats): k_fc0_x3
        m_full, f_full = 4 * hw * M_X3_HALF, 4 * hw * F_X3_HALF
    m_win, f_win = (2 * M_X3_HALF * 2, 2 * F_X3_HALF * 2) if fcode_all == 1 else (2 * M_MX_STEP, 2 * F_MX_STEP)  # per window pixel and tile (two super-steps; f16 format: four half-steps)
    mfma_fc0 = wk["work_win_pixels"] * m_win + (wk["work_full_tiles"] + dense_tiles) * m_full
    flop_exec_fc0 = wk["work_win_pixels"] * f_win + (wk["work_full_tiles"] + dense_tiles) * f_full
    mfma_tail = rows / 128.0 * (M_FC1_TILE + M_HEADS_TILE)
    executed = {"k_trunk": (mfma_trunk, mfma_trunk * 32768.0), "k_fc0_mx": (mfma_fc0, flop_exec_fc0), "tail": (mfma_tail, mfma_tail * 32768.0)} if have_work else {}
    SIMD_CYCLES_PER_S = 256 * 4 * BOOST_SCLK_MHZ * 1e6

    def executed_fields(kernel, sec, flop_row):
        if kernel not in executed or sec <= 0:
            return {}
        m, f = executed[kernel]
        return {"executed_flops": f / launches, "executed_mfma": m / launches, "frac_executed": 32.0 * m / (sec * SIMD_CYCLES_PER_S),
                "executed_over_algorithmic": f / max(rows * flop_row, 1.0)}

    def mfma_roofline(kernel, flop_row, note):
        sec = k_ms[kernel] * 1e-3
        ach = rows * flop_row / sec / 1e12 if sec > 0 else 0.0
        per_row = pmc.get(kernel + "_hbm_bytes_per_row_calibrated") or pmc.get(kernel + "_hbm_bytes_per_row") or PMC_HBM_BYTES_PER_ROW.get(n, {}).get(kernel)
        per_row_x2 = pmc.get(kernel + "_hbm_bytes_per_row")
        # Bytes a launch has to move per request row.  N = 9: trunk = the 384-B-per-pixel operand row it writes, fc0 = that row read once.
        # N = 15 (difference path): trunk = the child's 49-pixel difference row written (18816 B) + the base's 49 entries read
        # + per run of ~15 siblings whose base is not cached (22 % of the runs over a configs[1] episode, OMOK_SIB_STATS) a full row written twice
        # (fc0's compact copy, the base slot) and the base's three h grids; fc0 = the difference row + that share of a full row.
        # (bytes per pixel of an operand row: 2 x 192 in the fp6 format, 2 x 256 in the f16 format; a difference row = 49 pixels)
        fcode = int(st.get("fc0_format", 0))  # 0 fp6, 1 f16, 2 mixed (f16 full rows, fp6 difference rows)
        ppx = 512.0 if fcode in (1, 2) else 384.0
        drow = 49.0 * (512.0 if fcode == 1 else 384.0)
        miss, run = (0.22, 15.0) if n == 15 else (0.25, 7.5)  # share of the runs whose base is evaluated in full, siblings per run
        # (k_sib_children2, the default: a base slot also holds the base's d grids and its residual stream in front of block 2 -- 1280 B per pixel instead of
        #  384 -- written with the base and read once per run by its children; OMOK_SIB_V2=0: k_sib_children, 3 h grids)
        slot_px = 1280.0 if st.get("children2_launches", 0.0) > 0 else 384.0
        # Trunk group, per request row: the child's difference row written (drow); of its base it reads the 49 operand entries of its window (drow again), the h / d
        # grids of blocks 1 and 2 and the d grid of block 0 on its 25-pixel tile (640 B per pixel), the d grid of block 2 and the residual stream in front of block 2 on the
        # 24 ring pixels (640 B per pixel): 2 drow + 49 x 640 B = 69.0 KB in the fp6 format -- what the kernel moves when nothing is shared between siblings (`design`).
        # A run's ~15 siblings have windows all over the board, so the least a run can read is its base slot once (1280 B per pixel + the operand row): that / run + drow
        # = 43.8 KB per row (`alg_row`, the figure the roofline uses).  Both + the share of base passes (runs whose base is not cached).
        base_pass = miss * (2 * ppx * hw + slot_px * hw) / run
        design_row = {"k_trunk": 2 * drow + 49.0 * 640.0 + base_pass if slot_px > 384.0 else 2 * drow + 49.0 * 384.0 + base_pass, "k_fc0_mx": None}[kernel]
        alg_row = {"k_trunk": drow + (slot_px * hw + ppx * hw) / run + base_pass, "k_fc0_mx": drow + miss * ppx * hw / run + 2048}[kernel]
        if n != 15:  # (N = 9 rounds below the difference path's threshold and the f32 mode write / read whole operand rows)
            design_row = None
        members = {"k_trunk": "k_sib_children2 (k_sib_children on the copy path) + k_trunk<BASE> + k_trunk<rows> + k_group + k_bin_prefix (search rounds); k_trunk otherwise",
                   "k_fc0_mx": "k_fc0_mx | k_fc0_x3 <full rows, split-K> + k_facc_reduce + <window tiles> + k_win_finish (search rounds on the difference path); dense k_fc0_mx | k_fc0_x3 (+ k_splitk_finish) otherwise"}[kernel]
        # which roof is lower for this kernel group: time per row at the MFMA peak vs time per row at the HBM peak for the bytes it has to move
        alg_launch = alg_row * rows / launches + ({"k_trunk": 110e3, "k_fc0_mx": 128.0 * hw * 512 * 3}[kernel])
        t_mfma, t_hbm = flop_row / (F16_DENSE_PEAK_TFLOPS * 1e12), alg_row / (HBM_PEAK_GBS * 1e9)
        hbm_ach = alg_launch * launches / sec / 1e9 if sec > 0 else 0.0
        sides = {"mfma": {"achieved": ach, "peak": F16_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / F16_DENSE_PEAK_TFLOPS, "ns_per_row_at_peak": 1e9 * t_mfma},
                 "hbm": {"achieved": hbm_ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm_ach / HBM_PEAK_GBS, "ns_per_row_at_peak": 1e9 * t_hbm}}
        # the net's roof is MFMA on the algorithmic flops of a full evaluation.  (Round 4's line reported the trunk group against an HBM roof whose "algorithmic
        # bytes" were built from this design's own base-slot layout -- a bigger slot raised the figure.  That number now lives under design_traffic, named for what it is.)
        bound = "mfma"
        # What has to cross HBM per request row whatever the slot layout: the row the group hands to fc0 (N = 15: the 49-pixel difference row) + per run of siblings the
        # base's operand row once; fc0: that difference row read + per run the base row's share.  traffic / needed = the re-reads and layout overhead the counters see.
        needed_row = {"k_trunk": drow + ppx * hw / run, "k_fc0_mx": drow + miss * ppx * hw / run}[kernel] if n == 15 else ppx * hw
        return {"bound": bound, "kernel": kernel, "kernel_members": members, "achieved": sides[bound]["achieved"], "peak": sides[bound]["peak"], "unit": sides[bound]["unit"],
                "frac": sides[bound]["frac"], **executed_fields(kernel, sec, flop_row),
                "algorithmic_credit": "achieved = request rows x the 2*MAC flops of a FULL evaluation of this stage (SURVEY 8d) / HIP-event time of the group: at N = 15 sibling requests "
                                      "are evaluated as one base position + per-child window differences, so ~0.2-0.3x of those flops are executed (as 3 split-operand MFMAs per product); "
                                      "mfma_busy_pmc is the matrix pipes' measured busy fraction",
                "needed_bytes_per_row": needed_row,
                "traffic_over_needed": (per_row / needed_row) if per_row else None,
                "design_traffic": {"what": "HBM-side view of the same group: bytes this design moves per launch by its own data layout (NOT algorithmic bytes: they grow with the base-slot layout)",
                                   "bytes_per_launch_shared_bases": alg_launch, "bytes_per_launch_no_sharing": design_row * rows / launches if design_row else None,
                                   "gbs_at_shared_bases": hbm_ach, "frac_of_hbm_peak": hbm_ach / HBM_PEAK_GBS, "hbm_peak_gbs": HBM_PEAK_GBS},
                "both_roofs": sides,
                "traffic": per_row * rows / launches if per_row else None,
                "traffic_uncalibrated_x2": per_row_x2 * rows / launches if per_row_x2 else None,
                "traffic_unit": "HBM bytes per (average) launch: per-row bytes of the committed rocprofv3 --pmc pass (profiles/pmc_bytes.json: "
                                "FETCH_SIZE x 2 + WRITE_SIZE; fc0 calibrated for its 64-B residual requests, the flat x2 figure beside it) x rows per launch",
                "avg_launch_ms": k_ms[kernel] / launches, "rows_per_launch": rows / launches, "share_of_kernel_time": k_ms[kernel] / max(sum(k_ms.values()), 1e-9),
                "mfma_busy_pmc": pmc.get(kernel + "_mfma_busy"), "valu_busy_pmc": pmc.get(kernel + "_valu_busy"), "lds_busy_pmc": pmc.get(kernel + "_lds_busy"),
                "busy_pmc_unit": "fraction of the kernel group's cycles its SIMDs' matrix pipes / vector ALUs / the CUs' LDS were busy, from the committed "
                                 "rocprofv3 --pmc passes of this workload (profiles/pmc_bytes.json): SQ_VALU_MFMA_BUSY_CYCLES, 4 x SQ_INSTS_VALU, SQ_LDS_IDX_ACTIVE "
                                 "over GRBM_GUI_ACTIVE",
                "note": note}

    note_trunk = ("`kernel` = the trunk GROUP of a forward: conv_in + 3 bottleneck blocks (k_group + k_bin_prefix + k_trunk<BASE> + k_sib_children2 + k_trunk on the "
                  "rows outside sibling runs); avg_launch_ms = the group per forward, averaged over ALL rounds of the timed region (thin ones included; "
                  "`--max-plies 4` gives the full-round figure that profiles/r04_rocprofv3_kernel_stats_c2_first4plies.csv sums to).  "
                  "ALGORITHMIC flops 2*MAC of a full evaluation (13.0 MFLOP/eval at N = 15) / HIP-event time on the engine's stream.  Every product "
                  "runs as 3 f16 MFMAs (split operands), and at N = 15 sibling requests share a base pass and recompute only a 5x5 / 7x7 window each, so "
                  "the EXECUTED matrix work is ~0.2x the algorithmic figure: the mfma side of `both_roofs` counts useful work; the group's own unit utilisation is in "
                  "mfma_busy_pmc / valu_busy_pmc / lds_busy_pmc.  HBM side: design_traffic / traffic / needed_bytes_per_row; measured (profiles/r04_children_traffic_experiments.txt, "
                  "profiles/r05_children_store_order.txt): with every base read an L2 hit the group is 13 % faster, without its store instructions 17 % (with the same stores "
                  "kept in L2: no change -- the instructions, not the HBM writes) -- the rest is the dependent instruction chain of two waves per SIMD")
    fcode = int(st.get("fc0_format", 0))
    mix = {0: "Per K = 64 the kernel (k_fc0_mx) issues 4 f16 + 2 block-scaled fp6 MFMAs (split operands) = 1.5x the pipe time of a plain-f16 product (frac <= 0.67 for a dense fc0)",
           1: "f16 operand format (k_fc0_x3): 3 f16 MFMAs per product = 3x the pipe time of a plain-f16 product (frac <= 0.33 for a dense fc0)",
           2: "mixed operand format: the window tiles of the difference rows -- where the time goes -- run on k_fc0_mx (4 f16 + 2 block-scaled fp6 MFMAs per K = 64, 1.5x the pipe time of a "
              "plain-f16 product), full rows (one per run of siblings whose base is not cached, thin rounds) on k_fc0_x3 (3 f16 MFMAs per product)"}.get(fcode, "")
    note_fc0 = ("ALGORITHMIC flops 2*128*HW*512 per eval / HIP-event time of all fc0 launches.  " + mix + ".  A search round "
                "runs the dense fc0 only on one full row per run of siblings whose base is not cached and 98 of the 2*HW K-steps (the 7x7 window) on each "
                "child's difference row, i.e. ~0.28x (N = 15) / ~0.65x (N = 9) of the algorithmic work is EXECUTED: `achieved` counts useful work and can exceed "
                "what a dense kernel could reach")
    fc0_fmt = {0: "block-scaled fp6 (e2m3)", 1: "f16", 2: "mixed: f16 on full operand rows, block-scaled fp6 on the difference rows of sibling rounds"}.get(int(st.get("fc0_format", 0)), "f32")
    fc0_short = {0: "fp6", 1: "f16", 2: "mixed f16/fp6"}.get(int(st.get("fc0_format", 0)), "f32")
    net_s = (st["ms_trunk"] + st["ms_fc0"] + st["ms_tail"]) * 1e-3
    tree_s = st["ms_tree"] * 1e-3
    tree_traffic = pmc.get("tree_hbm_bytes_per_sim")
    out = {
        "metric": "self-play games/sec (15x15, 800 sims/move); MCTS nodes/sec",
        "value": games_per_s, "unit": "games/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / max(args.steps, 1), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": f"f16x3 split MFMA, fp32 acc; fc0 corrections {fc0_short}" if args.net_mode != "f32" else "f32",
        "dtype_long": (f"f16 (split hi+lo MFMA operands: f16 main term + two correction terms, fp32 accumulate; fc0's correction terms in {fc0_fmt}, "
                       + ("chosen by omok_net_commit's probe)" if args.net_mode == "f16x3" else "forced by --net-mode)")) if args.net_mode != "f32" else "f32",
        "fc0_format": {"in_use": fc0_fmt, "in_use_short": fc0_short, "probe_rows": st.get("probe_rows"), "probe_limit": st.get("probe_limit"),
                       "fp6_max_dp_dv": [st.get("probe_dp_fp6"), st.get("probe_dv_fp6")], "f16_max_dp_dv": [st.get("probe_dp_f16"), st.get("probe_dv_f16")],
                       "probe_logit_abs_max": st.get("probe_logit_max"),
                       "plain_rows_max_dlogit": {"fp6": st.get("probe_dlogit_fp6"), "f16": st.get("probe_dlogit_f16")}, "probe_logit_limit": st.get("probe_logit_limit"),
                       "probe_verdict": {"code": st.get("probe_outside"),
                                         "meaning": "0 = every figure of the committed format inside the margin limits; 1 = outside the margin (3e-4 on p / v, 5e-4 on the logits) but inside "
                                                    "north_star's 1e-3: committed, one line on stderr; 2 = the f16 format itself is outside 1e-3: the engine runs the fp32 kernels"},
                       "sibling_round": {"rows_checked": st.get("probe_round_rows"),
                                         "fp6_dp_dv_dlogit": [st.get("probe_round_dp_fp6"), st.get("probe_round_dv_fp6"), st.get("probe_round_dlogit_fp6")],
                                         "mixed_dp_dv_dlogit": [st.get("probe_round_dp_mixed"), st.get("probe_round_dv_mixed"), st.get("probe_round_dlogit_mixed")],
                                         "f16_dp_dv_dlogit": [st.get("probe_round_dp_f16"), st.get("probe_round_dv_f16"), st.get("probe_round_dlogit_f16")],
                                         "what": "one synthetic round of sibling runs through the difference path (base rows + 7x7-window difference rows) in each format, "
                                                 "against the fp32 kernels"},
                       "rule": "omok_net_commit keeps the fastest of fp6 < mixed (f16 full rows, fp6 difference rows) < f16 whose probe figures are inside the limits "
                               "(|dp|, |dv| <= probe_limit = 3e-4, |dlogit| <= probe_logit_limit = 5e-4 against the fp32 kernels): fp6 on the plain rows AND the synthetic "
                               "sibling round, mixed on the sibling round (its full rows are the f16 format's); f16 otherwise (DESIGN 3.4)"} if args.net_mode != "f32" else None,
        "children_kernel_launches": {"k_sib_children2": st.get("children2_launches"), "k_sib_children": st.get("children1_launches"),
                                     "note": "sibling rounds of the timed region by the kernel that evaluated the runs' children: k_sib_children2 on the difference path "
                                             "(rounds of >= 2048 rows at N = 15 in the mixed operand format, >= 3072 in the fp6 / f16 formats, >= 1024 at N = 9), k_sib_children on the copy path (smaller rounds)"},
        "data": "synthetic (empty boards, random-init net seed 0)",
        "config": {"workload": f"{games} concurrent {n}x{n} games per GPU, {args.sims} sims/move, K={k}, two trees per game"
                               + ("" if complete else f", first {args.max_plies} plies only (games/s extrapolated)"),
                   "games_per_gpu": games, "board": n, "sims_per_move": args.sims, "batch_k": k,
                   "warmup_step": f"episode cut after {args.warmup_plies} plies" if args.warmup_plies else "whole episode",
                   "parallelism": f"games sharded x{world}, no hot-path collective" + (f", {'RCCL' if backend == 'nccl' else backend} all-gather-v of replay tuples per episode" if args.gather else "")},
        "mcts_sims_per_s": sims / dt, "nn_evals_per_s": evals / dt, "plies_per_s": ply_games / dt,
        "mean_plies_per_game": mean_plies, "games_finished": finished,
        "roofline": mfma_roofline(dominant, flop_trunk if dominant == "k_trunk" else flop_fc0, note_trunk if dominant == "k_trunk" else note_fc0),
        "roofline_trunk": mfma_roofline("k_trunk", flop_trunk, note_trunk),
        "roofline_fc0": mfma_roofline("k_fc0_mx", flop_fc0, note_fc0),
        "roofline_net": {"bound": "mfma", "achieved": rows * flop_eval / net_s / 1e12 if net_s > 0 else 0.0, "peak": F16_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": (rows * flop_eval / net_s / 1e12 / F16_DENSE_PEAK_TFLOPS) if net_s > 0 else 0.0,
                         **({"executed_flops": sum(v[1] for v in executed.values()) / launches, "executed_mfma": sum(v[0] for v in executed.values()) / launches,
                             "frac_executed": 32.0 * sum(v[0] for v in executed.values()) / (net_s * SIMD_CYCLES_PER_S)} if executed and net_s > 0 else {}),
                         "note": "whole forward per SURVEY 8d: evals x 43.25 MFLOP / (trunk + fc0 + tail time)"},
        "roofline_tree": {"bound": "hbm", "kernel": "k_round+k_scan+k_scatter", "achieved": st["tree_bytes"] / tree_s / 1e9 if tree_s > 0 else 0.0,
                          "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": (st["tree_bytes"] / tree_s / 1e9 / HBM_PEAK_GBS) if tree_s > 0 else 0.0,
                          "algorithmic_bytes_per_sim": st["tree_bytes"] / max(st["sims"], 1.0),
                          "traffic": tree_traffic * st["sims"] / max(st["round_launches"], 1.0) if tree_traffic else None,
                          "hbm_gbs_pmc": tree_traffic * st["sims"] / tree_s / 1e9 if tree_traffic and tree_s > 0 else None,
                          "hbm_bytes_per_sim_pmc": tree_traffic,
                          "on_chip_fraction": (max(0.0, 1.0 - tree_traffic / (st["tree_bytes"] / max(st["sims"], 1.0))) if tree_traffic and st["tree_bytes"] > 0 else None),
                          "on_chip_note": "SURVEY 8d asks for the LDS-resident fraction of the tree: this design stages NO tree array in LDS -- what a round reuses is one leaf per tree, "
                                          "kept in registers (DESIGN 3, 'Where the tree lives') -- so the figure reported is the share of the kernel-counted algorithmic bytes that never "
                                          "reach HBM by the PMC counters (registers + L2): 1 - hbm_bytes_per_sim_pmc / algorithmic_bytes_per_sim, 0 if the counters see more than the algorithm needs",
                          "traffic_unit": "HBM bytes per round (k_round + k_scan + k_fill + k_scatter*): PMC bytes per simulation (profiles/) x simulations per round"},
        "rank0_kernel_ms": {kk: st[kk] for kk in ("ms_round", "ms_tree", "ms_trunk", "ms_fc0", "ms_tail", "ms_ply")},
        "rank0_kernel_ms_method": f"HIP events on the engine's stream around every kernel category of 1 search round in {max(1, args.profile_every)} "
                                  "(ply-level work: always), sampled sums scaled by rounds seen / rounds timed",
        "rank0_timed_region_ms": 1e3 * dt,
        "game_length_percentiles": {str(q): float(np.percentile(plies, q)) for q in (0, 10, 25, 50, 75, 90, 99, 100)},
        "arena": {"max_nodes": max_nodes, "max_tables": max_tables, "peak_nodes": st["peak_nodes"], "peak_tables": st["peak_tables"]},
        "cpu_baseline": None,
        "clocks": clocks,
        "seconds_since_process_start": elapsed(),
    }
    if clocks and clocks["sclk_mhz_busy_median"] > 0:
        # `peak` is quoted at the boost clock (2400 MHz); the well-filled rounds run at the package power limit, below it: the same fraction against the peak at the clock they got
        scale = BOOST_SCLK_MHZ / clocks["sclk_mhz_busy_median"]
        for key in ("roofline", "roofline_trunk", "roofline_fc0", "roofline_net"):
            if out[key]["frac"] * scale < 1.0:  # (an algorithmic-credit figure above 1 is not a fraction of anything: fc0's executes ~0.25x of its credit)
                out[key]["frac_at_sustained_clock"] = out[key]["frac"] * scale
            if out[key].get("frac_executed") is not None:
                out[key]["frac_executed_at_sustained_clock"] = out[key]["frac_executed"] * scale
        clocks["boost_sclk_mhz"] = BOOST_SCLK_MHZ
    if args.gather:
        out["replay_gather"] = {"records_per_episode": gathered["records"] / max(args.steps, 1), "bytes_per_episode": gathered["bytes"] / max(args.steps, 1),
                                "seconds_per_episode": gathered["seconds"] / max(args.steps, 1), "inside_timed_region": True,
                                "method": "8 x int64 counts all-gather, then exact-size grouped send/recv (all-gather-v)", "backend": "RCCL" if backend == "nccl" else backend,
                                "last_counts": gathered.get("last_counts"), "last_ids": gathered.get("last_ids")}
    emit(out, final=False)  # the measured line exists from here on, whatever happens to the extra legs

    # ---- extra legs, outside the timed region, only while the wall-clock budget has room ----------------------------
    extras = False

    def room(need):
        return elapsed() + need < args.budget_seconds

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    for leg_name, leg_mode in (("f16", "NET_F16X3_F16"), ("fp6", "NET_F16X3_FP6")):
        # The other operand formats on the same workload, one more whole episode each with its own engine: f16 everywhere (the most precise: three f16 MFMAs per
        # product of fc0) and fp6 everywhere (the fastest; its LOGITS miss north_star's 1e-3 on this net, which is why omok_net_commit's probe does not keep it).
        if not (use_cuda and world == 1 and complete and args.f16_leg and args.net_mode == "f16x3" and room(30 + 1.6 * dt / max(args.steps, 1))):
            break
        if B.FC0_FORMATS.get(int(st.get("fc0_format", -1))) == leg_name:
            continue  # (the headline already ran in this format)
        try:
            eng2 = oa.Engine(board_size=n, games=games, max_nodes=max_nodes, max_tables=max_tables, max_batch_k=k, device=gpu, net_mode=getattr(B, leg_mode),
                             seed=args.seed, game_offset=oa.dist.game_offset(rank, games))
            eng2.load_random_weights(0)
            sp2 = oa.SelfPlay(eng2)
            sp2.reset()
            sp2.run(args.sims, k, 0.25, 0.03, 1.0, 30, args.warmup_plies or 4)  # warm-up: an episode cut after a few plies
            sp2.set_episode(1)
            eng2.reset_stats()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            sp2.reset()
            sp2.run(args.sims, k, 0.25, 0.03, 1.0, 30, 0)
            torch.cuda.synchronize()
            dt2 = time.perf_counter() - t1
            st2 = eng2.stats()
            out[f"value_{leg_name}_format"] = st2["finished"] / dt2
            out[f"{leg_name}_format_leg"] = {"games_per_s": st2["finished"] / dt2, "seconds": dt2, "episodes": 1, "games_finished": st2["finished"],
                                             "mcts_sims_per_s": st2["sims"] / dt2, "fc0_format": B.FC0_FORMATS[int(st2["fc0_format"])],
                                             "ratio_to_value": st2["finished"] / dt2 / max(games_per_s, 1e-9),
                                             "note": "extra leg outside the timed region: the same workload (one whole episode after a cut warm-up episode, its own engine) with "
                                                     f"fc0's operand format forced ({leg_mode})"}
            eng2.close()
            del sp2, eng2
        except Exception as ex:
            out[f"{leg_name}_format_leg"] = {"error": repr(ex)}
        extras = True
    if use_cuda and world == 1 and complete:
        if args.precision_rows > 0 and room(25):
            try:
                out["precision"] = precision_check(args, args.precision_rows, gpu)
            except Exception as ex:
                out["precision"] = {"error": repr(ex)}
            extras = True
        if room(15):  # the episode-end replay post-processing row on the device
            try:
                n_rec = 6 * int(plies.sum())
                buf = torch.empty(max(n_rec, 1) * rec, dtype=torch.uint8, device=device)
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
                if args.train_steps > 0 and room(30):  # the training phase on the same records
                    from omok_ai_amd import train as T
                    ph = T.TrainPhase(n, oa.weights.init_random(n, seed=0), device)
                    ph.run(buf, update_count=2, batch_size=128, seed=0)  # warm-up (MIOpen / rocBLAS plans)
                    torch.cuda.synchronize()
                    t2 = time.perf_counter()
                    v_l, p_l, l_ = ph.run(buf, update_count=args.train_steps, batch_size=128, seed=1)
                    torch.cuda.synchronize()
                    dt_tr = time.perf_counter() - t2
                    out["train_phase"] = {"steps": args.train_steps, "batch": 128, "steps_per_s": args.train_steps / dt_tr, "loss": l_,
                                          "note": "AgentModel::train (Adadelta lr 0.01) via torch autograd on the augmented replay records, rank 0"}
                del buf
            except Exception as ex:  # an extra line of the report must never cost the bench line itself
                out["replay_postprocess_error"] = repr(ex)
            extras = True
        if args.window_plies > 0 and room(8 + 0.5 * dt / max(args.steps, 1)):
            # a fixed-length window (the first 20 plies: every game is still alive, every round is full) for run-to-run stability
            try:
                sp.set_episode(args.warmup + args.steps)
                sp.reset()
                eng.set_profiling(0)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                wst = sp.run(args.sims, k, 0.25, 0.03, 1.0, 30, args.window_plies)
                torch.cuda.synchronize()
                dt_w = time.perf_counter() - t1
                rounds_up = (args.sims + k - 1) // k * k
                out["window_first_plies"] = {"plies": args.window_plies, "seconds": dt_w, "ply_games_per_s": games * args.window_plies / dt_w,
                                             "mcts_sims_per_s": games * args.window_plies * rounds_up / dt_w, "ms_per_full_round": 1e3 * dt_w / (args.window_plies * rounds_up / k),
                                             "games_per_s_at_mean_length": games * args.window_plies / dt_w / max(mean_plies, 1.0),
                                             "note": "untimed-region extra leg, no profiling events: the same engine plays the first plies of one more episode "
                                                     "(all games alive, all rounds full); games_per_s_at_mean_length = what the episode would reach if every round were full"}
                eng.set_profiling(max(1, args.profile_every))
            except Exception as ex:
                out["window_first_plies"] = {"error": repr(ex)}
            extras = True
        if args.slots_multiple > 0 and args.max_plies == 0 and hasattr(sp, "run_slots") and room(20 + 1.2 * args.slots_multiple * dt / max(args.steps, 1)):
            # Slots mode: the same games (by index) as an episode of slots_multiple x games, but a slot whose game is over takes the next
            # index instead of idling until the episode's longest game ends (the last 40 % of an episode's rounds hold < 10 % of its rows).  (After the post-processing leg: it resets the engine.)
            try:
                total = args.slots_multiple * games
                cap = int(total * min(hw, 1.25 * mean_plies + 8))
                buf = torch.empty(cap * rec, dtype=torch.uint8, device=device)
                sp.set_episode(args.warmup + args.steps + 1)
                sp.reset()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                sst, nrec, _, ln, _ = sp.run_slots(total, args.sims, k, buf.data_ptr(), cap, 0.25, 0.03, 1.0, 30)
                torch.cuda.synchronize()
                dt_s = time.perf_counter() - t1
                out["slots_mode"] = {"games": total, "slots": games, "seconds": dt_s, "games_per_s": total / dt_s, "records_packed": nrec,
                                     "mean_plies_per_game": float(ln.mean()),
                                     "note": "omok_selfplay_run_slots: finished slots restart with the next game index (RNG keyed by game index and the "
                                             "game's own ply: per-game results are those of one episode of all the games, tests/test_gpu_slots.py); "
                                             "includes packing every finished game's transitions; one call, wall time, rank 0; NOT `value` "
                                             "(the reference's schedule is the episode: its batch shrinks as games end)"}
                del buf
            except Exception as ex:
                out["slots_mode"] = {"error": repr(ex)}
            extras = True
    if args.cpu_seconds > 0 and world == 1:
        share = min(args.cpu_seconds, args.budget_seconds - elapsed() - 10.0)
        if share >= 8.0:
            try:
                out["cpu_baseline"] = cpu_baseline(args, max(mean_plies, 1.0), share)
            except Exception as ex:  # (the GPU measurement above stands on its own)
                out["cpu_baseline"] = {"error": repr(ex)}
            chk = out["cpu_baseline"].pop("_net_check", None)
            if chk is not None and use_cuda:
                # The GPU's outputs against the ORACLE (oracle/net.c), not against the GPU's own fp32 kernels.
                #  (1) difference_path: request rows of REAL search rounds of an engine of the timed engine's size, whose rounds take the path the headline is timed on (65536 rows per round at configs[1]:
                #      sibling base + 7x7-window difference rows, the operand format this engine's commit probe chose) -- inputs from omok_round_inputs, outputs from
                #      omok_round_outputs / omok_round_logits of those very rounds, >= 512 of the rows through the oracle's forward.  Round 4's line called the plain-row
                #      figure below "headline_mode"; a 64-game engine never takes the difference path, so that was a statement about plain rows only.
                #  (2) plain_rows: omok_evaluate_pv / omok_evaluate_logits (full operand rows: mirror evaluations, single rows, thin rounds) in the same mode and in the f16 format.
                try:
                    from omok_ai_amd import precision as PR
                    from oracle import oracle as O
                    mode_h = {"f16x3": B.NET_F16X3, "fp6": B.NET_F16X3_FP6, "mixed": B.NET_F16X3_MIXED, "f16": B.NET_F16X3_F16, "f32": B.NET_F32}[args.net_mode]
                    tensors0 = oa.weights.init_random(n, seed=0)
                    vs = {"reference": "oracle/net.c (fp32 restatement of network.rs, the checker of the -m gpu tests)"}
                    g_dp = max(PR.difference_path_games(n, k), -(-4096 // k), games)  # (the timed engine's own size: 65536-row rounds at configs[1])
                    rr = PR.search_round_rows(tensors0, n, games=g_dp, batch_k=k, warm_plies=3, warm_sims=64, rounds=2, device=gpu, seed=args.seed + 5, net_mode=mode_h)
                    sel = np.arange(0, len(rr["x"]), max(1, len(rr["x"]) // 768))[:768]
                    onet = O.Net(n, tensors0)
                    po, vo, lgo, vpo = onet.forward_logits(np.ascontiguousarray(rr["x"][sel]), threads=chk["threads"])
                    vs["difference_path"] = {"fc0_format": rr["fc0_format"], "rows_per_round": rr["rows_per_round"], "difference_path_rounds": rr["difference_path_rounds"],
                                             "rows_compared": int(len(sel)), "max_dp": float(np.abs(rr["p"][sel][:, :hw] - po).max()), "max_dv": float(np.abs(rr["v"][sel] - vo).max()),
                                             "max_dlogit": float(np.abs(rr["logits"][sel][:, :hw] - lgo).max()), "max_dvpre": float(np.abs(rr["vpre"][sel] - vpo).max()),
                                             "logit_abs_max": float(np.abs(lgo).max()),
                                             "what": "rows and outputs of real search rounds (omok_round_inputs / omok_round_outputs / omok_round_logits) of an engine in the headline's net mode "
                                                     f"with {g_dp} games x K = {k}: the rounds take the difference path (difference_path_rounds = launches of k_sib_children2 among them)"}
                    vs["headline_mode"] = vs["difference_path"]  # (the name round 4's line used -- now the path the headline runs)
                    vs["plain_rows"] = {"rows": int(len(chk["x"])), "what": "omok_evaluate_pv / omok_evaluate_logits on request rows of the CPU legs' search rounds (full operand rows)"}
                    for tag, mode in (("headline_mode", mode_h), ("f16_format", B.NET_F16X3_F16)):
                        e3 = oa.Engine(board_size=n, games=64, max_nodes=8, max_tables=4, max_batch_k=k, device=gpu, net_mode=mode)
                        e3.load_random_weights(0)
                        pg, vg = e3.evaluate_pv(chk["x"])
                        lg, vpg = e3.evaluate_logits(chk["x"])
                        fmt3 = B.FC0_FORMATS[int(e3.stats()["fc0_format"])]
                        e3.close()
                        vs["plain_rows"][tag] = {"fc0_format": fmt3, "max_dp": float(np.abs(pg.reshape(len(chk["x"]), -1) - chk["p"]).max()),
                                                 "max_dv": float(np.abs(vg.reshape(-1) - chk["v"]).max()),
                                                 "max_dlogit": float(np.abs(lg.reshape(len(chk["x"]), -1)[:, :hw] - chk["logits"]).max()),
                                                 "max_dvpre": float(np.abs(vpg.reshape(-1) - chk["vpre"]).max())}
                    vs["plain_rows"]["logit_abs_max"] = float(np.abs(chk["logits"]).max())
                    vs["north_star_logits_1e-3"] = {"difference_path": bool(vs["difference_path"]["max_dlogit"] < 1e-3 and vs["difference_path"]["max_dvpre"] < 1e-3
                                                                            and vs["difference_path"]["difference_path_rounds"] > 0),
                                                    **{"plain_rows_" + tag: bool(vs["plain_rows"][tag]["max_dlogit"] < 1e-3 and vs["plain_rows"][tag]["max_dvpre"] < 1e-3)
                                                       for tag in ("headline_mode", "f16_format")}}
                    out.setdefault("precision", {})["vs_oracle"] = vs
                except Exception as ex:
                    out.setdefault("precision", {})["vs_oracle"] = {"error": repr(ex)}
        else:
            out["cpu_baseline"] = {"skipped": f"no room in the {args.budget_seconds:.0f} s budget ({elapsed():.0f} s used)"}
        extras = True
    out["seconds_since_process_start"] = elapsed()
    emit(out, final=True)  # the LAST line is the complete one (also when no extra leg ran: it then differs from the first only by the keys that are null)


if __name__ == "__main__":
    main()
