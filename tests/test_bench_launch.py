"""`python bench.py --gpus N` as the driver runs it (no torchrun around it), on the CPU: the parent starts N ranks by itself,
every rank gets its shard (game_offset = rank * games), the replay gather ships exact (uneven) record counts, and rank 0
prints a result line with n_gpus = N.  The engine is replaced by tests/mock/mock_engine.py (no GPU here); the backend is
gloo.  On a GPU box the same code path runs with the real engine over RCCL."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, tmp_path, extra_env=None):
    env = dict(os.environ)
    env.update(OMOK_BENCH_ENGINE="mock.mock_engine", OMOK_BENCH_BACKEND="gloo", OMOK_MOCK_DIR=str(tmp_path),
               PYTHONPATH=os.pathsep.join([ROOT, os.path.join(ROOT, "tests")] + env.get("PYTHONPATH", "").split(os.pathsep)))
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    env.update(extra_env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=300)


def test_bench_self_launches_two_ranks_and_gathers_exact_counts(tmp_path):
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--games", "6", "--gather", "--cpu-seconds", "0"], tmp_path)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [json.loads(x) for x in r.stdout.splitlines() if x.startswith("{")]
    assert lines, r.stdout + r.stderr[-2000:]
    out = lines[-1]
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert out["games_finished"] == 2 * 2 * 6  # both ranks' games, both steps (sum over ranks)
    for key in ("metric", "value", "unit", "ms_per_step", "higher_is_better", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in out, key
    ranks = [json.load(open(tmp_path / f"rank{i}.json")) for i in range(2)]
    assert [x["game_offset"] for x in ranks] == [0, 6] and all(x["world"] == "2" for x in ranks)
    g = out["replay_gather"]
    assert g["last_counts"] == [6, 12]            # uneven live counts, no fixed-capacity slabs
    assert g["last_ids"] == list(range(6)) + [6 + i % 6 for i in range(12)]  # rank order = global game order
    assert g["bytes_per_episode"] == 18 * (228 + 900 + 4)


def test_bench_self_launches_eight_ranks(tmp_path):
    """the driver's N = 8 command line (`python bench.py --gpus 8 ...`) end to end on the CPU: eight ranks, eight shards, the replay gather
    with eight uneven counts, one result line for the whole job"""
    r = _run(["--gpus", "8", "--steps", "1", "--warmup", "1", "--games", "4", "--gather", "--cpu-seconds", "0"], tmp_path, {"OMP_NUM_THREADS": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    out = [json.loads(x) for x in r.stdout.splitlines() if x.startswith("{")][-1]
    assert out["n_gpus"] == 8 and out["scaling"] == "weak" and out["games_finished"] == 8 * 4
    assert out["config"]["parallelism"].startswith("games sharded x8")
    ranks = [json.load(open(tmp_path / f"rank{i}.json")) for i in range(8)]
    assert [x["game_offset"] for x in ranks] == [4 * i for i in range(8)] and all(x["world"] == "8" for x in ranks)
    g = out["replay_gather"]
    assert g["last_counts"] == [4 * (i + 1) for i in range(8)]
    assert g["last_ids"] == [4 * rk + i % 4 for rk in range(8) for i in range(4 * (rk + 1))]  # rank order = global game order


def test_bench_says_so_when_the_gpus_are_not_there(tmp_path):
    r = _run(["--gpus", "4"], tmp_path, {"OMOK_BENCH_BACKEND": "nccl"})  # nccl = count real devices: none in this container
    import torch
    if torch.cuda.device_count() >= 4:
        return
    assert r.returncode != 0  # nothing was measured: not a success
    out = json.loads(r.stdout.splitlines()[-1])
    assert "4 GPUs needed" in out["error"] and out["value"] is None


def test_result_line_is_short_and_complete(tmp_path):
    """The driver parses ONE line of a few KB (round 5's had grown to 24 KB and its cpu_baseline fell out of the record): the last line stays under 8000 characters
    and carries the contract's keys, `roofline` and `cpu_baseline` (here: the mock engine + the real CPU leg, bounded to a few seconds); the verbose record goes
    to the sidecar file."""
    full = tmp_path / "full.json"
    r = _run(["--steps", "1", "--warmup", "0", "--games", "4", "--cpu-seconds", "14"], tmp_path, {"OMOK_BENCH_FULL": str(full)})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [x for x in r.stdout.splitlines() if x.startswith("{")]
    assert len(lines) == 2 and all(len(x) < 8000 for x in lines)
    first, last = json.loads(lines[0]), json.loads(lines[-1])
    assert "cpu_baseline" not in first and not any(v is None for k, v in first.items() if k != "vs_baseline")  # the early safety line has no null placeholders
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in last, key
    assert len(last["dtype"]) <= 80 and last["notes"] == "profiles/bench_line_notes.md"
    cb = last["cpu_baseline"]
    assert cb["value"] > 0 and cb["cores"] >= 1 and cb["kind"] == "port" and cb["unit"] == "games/s" and len(cb["sample"]) < 200 and cb["net_tflops"] > 0
    rf = last["roofline"]
    assert rf["bound"] in ("hbm", "mfma") and {"achieved", "peak", "unit", "frac", "traffic"} <= set(rf)
    assert all(not isinstance(v, str) or len(v) <= 120 for v in rf.values())  # numbers and short identifiers only
    assert os.path.exists(os.path.join(ROOT, "profiles", "bench_line_notes.md"))
    rec = json.load(open(full))  # the verbose record of the same run
    assert rec["value"] == first["value"] or abs(rec["value"] - first["value"]) <= 1e-4 * abs(rec["value"])
    assert "legs" in rec["cpu_baseline"] and "mm_calibration_tflops_by_threads" in rec["cpu_baseline"]


def test_compact_line_of_a_real_verbose_record():
    """compact_line on round 5's real 24-KB line (profiles/r05_bench_c2_20steps_driver_style_final_build.json): under the limit, and the oracle precision leg, the
    CPU baseline and every roofline object survive as numbers."""
    import sys
    sys.path.insert(0, ROOT)
    import bench
    rec = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_c2_20steps_driver_style_final_build.json")))
    assert len(json.dumps(rec)) > 20000
    line = bench.compact_line(rec)
    text = json.dumps(line)
    assert len(text) < 8000, len(text)
    assert line["value"] == float(f"{rec['value']:.5g}") and line["cpu_baseline"]["cores"] == rec["cpu_baseline"]["cores"]
    vs = line["precision"]["vs_oracle"]
    assert vs["difference_path"]["max_dlogit"] < 1e-3 and vs["difference_path"]["difference_path_rounds"] > 0 and vs["north_star_logits_1e-3"]["difference_path"] is True
    for key in ("roofline", "roofline_fc0", "roofline_net", "roofline_tree"):
        assert 0 < line[key]["frac"] and "note" not in line[key] and "kernel_members" not in line[key]
