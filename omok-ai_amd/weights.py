"""Policy/value-net tensors in the reference's variable order, and the random initialiser.

Order and shapes follow alpha-zero/src/network.rs:78-79,113-122,149-150,162-163,201-202,240-241
(31 variables; conv kernels HWIO, fc weights [in, out]).  Initial values follow
network-utils/src/lib.rs:76-93: N(0,1) * scale with He = 2/sqrt(fan_in), Xavier =
2/sqrt(fan_in+fan_out); conv fan_in = kh*kw*cin (lib.rs:131-134); the depthwise kernel uses
He(kh*kw*cin) (lib.rs:192-195), the pointwise kernel He(cin) (lib.rs:223); biases are 0.
The reference draws from TF's unseeded RandomStandardNormal; this build draws from numpy's
seeded Generator so that every consumer of a (board size, seed) pair loads identical tensors.
"""
import numpy as np

C, M, F = 128, 32, 512
NUM_TENSORS = 31


def tensor_shapes(n):
    hw = n * n
    shapes = [(1, 1, 3, C), (C,)]
    for _ in range(3):
        shapes += [(1, 1, C, M), (M,), (3, 3, M, 1), (1, 1, M, M), (M,), (1, 1, M, C), (C,)]
    shapes += [(C * hw, F), (F,), (F, F), (F,), (F, 1), (1,), (F, hw), (hw,)]
    return shapes


def tensor_names():
    names = ["conv_w", "conv_b"]
    for i in range(3):
        names += [f"residual_{i}_conv0_w", f"residual_{i}_conv0_b", f"residual_{i}_conv1_w(depthwise)",
                  f"residual_{i}_conv1_w(pointwise)", f"residual_{i}_conv1_b", f"residual_{i}_conv2_w",
                  f"residual_{i}_conv2_b"]
    names += ["fc0_w", "fc0_b", "fc1_w", "fc1_b", "v_fc0_w", "v_fc0_b", "p_fc0_w", "p_fc0_b"]
    return names


def init_random(n, seed=0):
    """Random-init weights: list of 31 float32 arrays in reference order."""
    rng = np.random.default_rng(seed)
    hw = n * n
    he = lambda fan_in: np.float32(2.0) / np.sqrt(np.float32(fan_in))
    xavier = lambda fi, fo: np.float32(2.0) / np.sqrt(np.float32(fi + fo))
    scales = [he(3), None]
    for _ in range(3):
        scales += [he(C), None, he(9 * M), he(M), None, he(M), None]
    scales += [he(C * hw), None, he(F), None, xavier(F, 1), None, xavier(F, hw), None]
    out = []
    for shape, scale in zip(tensor_shapes(n), scales):
        if scale is None:
            out.append(np.zeros(shape, dtype=np.float32))
        else:
            out.append((rng.standard_normal(shape, dtype=np.float32) * np.float32(scale)).astype(np.float32))
    return out


def checksum(tensors):
    """Order-sensitive checksum used by the golden fixtures to detect generator drift."""
    acc = np.float64(0.0)
    for i, t in enumerate(tensors):
        acc += np.float64(i + 1) * np.abs(t.astype(np.float64)).sum()
    return float(acc)
