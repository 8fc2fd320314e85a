"""bench.py's clock / power sampler (host logic, no GPU): a child process that runs `rocm-smi` and is summarised over the timed region."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _fake_rocm_smi(tmp_path, body):
    exe = tmp_path / "rocm-smi"
    exe.write_text("#!/bin/sh\n" + body)
    exe.chmod(0o755)
    return str(tmp_path)


def test_sampler_summarises_the_busy_samples_of_the_timed_region(tmp_path, monkeypatch):
    import bench
    d = _fake_rocm_smi(tmp_path, 'echo "GPU[0]\t\t: sclk clock level: 3: (2001Mhz)"\necho "GPU[0]\t\t: Current Socket Graphics Package Power (W): 1290.0"\n'
                                 'echo "GPU[1]\t\t: sclk clock level: S: (94Mhz)"\necho "GPU[1]\t\t: Current Socket Graphics Package Power (W): 245.0"\n')
    monkeypatch.setenv("PATH", d + os.pathsep + os.environ["PATH"])
    monkeypatch.delenv("OMOK_BENCH_CLOCKS", raising=False)
    s = bench.start_clock_sampler()
    assert s is not None
    t0 = time.time()
    time.sleep(2.5)
    r = bench.stop_clock_sampler(s, t0, time.time())
    assert r is not None and r["samples"] >= 4
    assert r["sclk_mhz_busy_median"] == 2001.0 and r["power_w_busy_median"] == 1290.0  # the busiest GPU of each sample
    assert s[0].poll() is not None and not os.path.exists(s[1])  # the child has ended, its scratch file is gone


def test_sampler_stays_off_under_a_profiler_and_when_switched_off(tmp_path, monkeypatch):
    import bench
    d = _fake_rocm_smi(tmp_path, "exit 0\n")
    monkeypatch.setenv("PATH", d + os.pathsep + os.environ["PATH"])
    monkeypatch.setenv("OMOK_BENCH_CLOCKS", "0")
    assert bench.start_clock_sampler() is None
    monkeypatch.delenv("OMOK_BENCH_CLOCKS")
    monkeypatch.setenv("ROCPROF_COUNTERS", "pmc: FETCH_SIZE")  # what rocprofv3 --pmc exports: the GPU is initialised before the program starts
    assert bench.start_clock_sampler() is None
    monkeypatch.delenv("ROCPROF_COUNTERS")
    monkeypatch.setenv("LD_PRELOAD", "/opt/rocm/lib/rocprofiler-sdk/librocprofiler-sdk-tool.so")
    assert bench.start_clock_sampler() is None


def test_too_few_samples_give_no_clocks_object(tmp_path, monkeypatch):
    import bench
    d = _fake_rocm_smi(tmp_path, "exit 0\n")  # prints nothing: the sampler stops at once
    monkeypatch.setenv("PATH", d + os.pathsep + os.environ["PATH"])
    monkeypatch.delenv("OMOK_BENCH_CLOCKS", raising=False)
    monkeypatch.delenv("LD_PRELOAD", raising=False)
    s = bench.start_clock_sampler()
    t0 = time.time()
    time.sleep(0.5)
    assert bench.stop_clock_sampler(s, t0, time.time()) is None
