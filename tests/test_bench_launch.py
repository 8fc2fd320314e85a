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
    assert r.returncode == 0
    out = json.loads(r.stdout.splitlines()[-1])
    assert "4 GPUs needed" in out["error"] and out["value"] is None
