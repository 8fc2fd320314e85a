"""The C caller of the boundary (tests/c/harness.c, built by __graft_entry__.build()): what a Rust `extern "C"` user of
include/omok_mi355x.h does, checked with a C compiler since rustc is absent.  Without a GPU the harness can only take the
error path (omok_create -> OMOK_ERR_HIP + message, exit code 3); on the GPU box it plays two plies of two games."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HARNESS = os.path.join(ROOT, "tests", "c", "harness")


def _run():
    assert os.path.exists(HARNESS), "run __graft_entry__.build() first (make -C tests/c)"
    return subprocess.run([HARNESS], capture_output=True, text=True, timeout=600)


def test_c_caller_sees_a_clean_error_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: the gpu test runs the harness")
    out = _run()
    assert out.returncode == 3, (out.returncode, out.stdout, out.stderr)
    assert "OMOK_ERR_HIP" in out.stdout and "no CPU path" in out.stdout


@pytest.mark.gpu
def test_c_caller_plays_two_plies_through_the_abi():
    out = _run()
    assert out.returncode == 0, (out.returncode, out.stdout[-2000:], out.stderr[-2000:])
    assert "harness OK" in out.stdout
