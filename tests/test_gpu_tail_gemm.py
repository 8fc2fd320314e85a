"""GPU test of the tail GEMMs (fc1, value / policy heads: alpha-zero/src/network.rs:152-247) on whole-K launches: k_gemm_w (round 6: wave-private weight
rings, sample operands in chunks of two k-steps, one workgroup barrier per chunk) must return the bits of k_gemm_t (one barrier per k-step) -- every accumulator sums
the same products in the same order -- and both stay within north_star's 1e-3 of the oracle's forward.  Batches above 16384 rows take the whole-K launches
(forward_f16x3: tsplit == 1), smaller ones the K-split k_gemm_t launches, which are compared with the same rows of the big batch."""
import os

import numpy as np
import pytest

import omok_ai_amd as oa
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _inputs(n, rows, seed):
    rng = np.random.default_rng(seed)
    hw = n * n
    dens = rng.random((rows, 1)) * 0.6
    u = rng.random((rows, hw))
    board = np.where(u < dens / 2, 1, np.where(u < dens, 2, 0))
    x = np.zeros((rows, 3 * hw), dtype=np.float32)
    x[:, 0:2 * hw:2] = board == 1
    x[:, 1:2 * hw:2] = board == 2
    x[:, 2 * hw:] = rng.integers(0, 2, (rows, 1))
    return x


@pytest.mark.parametrize("n,games,rows", [(15, 1280, 20000), (9, 1280, 20321)])
def test_wave_private_tail_gemm_returns_the_bits_of_the_barrier_per_kstep_kernel(n, games, rows):
    tensors = oa.weights.init_random(n, seed=3)
    eng = oa.Engine(board_size=n, games=games, max_nodes=16, max_tables=8, max_batch_k=16)
    eng.load_weights(tensors)
    x = _inputs(n, rows, 11)
    old = os.environ.get("OMOK_GEMM_W")
    try:
        os.environ["OMOK_GEMM_W"] = "0"
        lg_t, vp_t = eng.evaluate_logits(x)
        p_t, v_t = eng.evaluate_pv(x)
        os.environ["OMOK_GEMM_W"] = "1"
        lg_w, vp_w = eng.evaluate_logits(x)
        p_w, v_w = eng.evaluate_pv(x)
    finally:
        if old is None:
            os.environ.pop("OMOK_GEMM_W", None)
        else:
            os.environ["OMOK_GEMM_W"] = old
    assert np.array_equal(lg_t, lg_w) and np.array_equal(vp_t, vp_w), "k_gemm_w logits differ from k_gemm_t's"
    assert np.array_equal(p_t, p_w) and np.array_equal(v_t, v_w)
    # a small batch of the same rows runs the K-split k_gemm_t launches: same rows, sums in another order -> close, not equal; and the oracle on a sample of rows
    sel = np.r_[0:64, rows - 64:rows]
    lg_s, vp_s = eng.evaluate_logits(x[sel])
    assert np.abs(lg_s - lg_w[sel]).max() < 2e-4 and np.abs(vp_s - vp_w[sel]).max() < 2e-4
    pc, vc = O.Net(n, tensors).forward(x[sel], threads=8)
    assert np.abs(p_w.reshape(rows, -1)[sel] - pc).max() < 1e-3 and np.abs(v_w.reshape(-1)[sel] - vc).max() < 1e-3
    eng.close()
