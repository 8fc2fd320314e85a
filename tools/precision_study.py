"""Precision study: which MFMA input dtype keeps p/v within 1e-3 of the fp32 path?

Emulates the policy/value net of the reference (alpha-zero/src/network.rs:51-262) in torch on
CPU, with random-init weights following WeightInitializer (network-utils/src/lib.rs:86-92),
and compares an fp64 evaluation against evaluations whose matmul inputs (weights and
activations) are rounded to fp16 / bf16 while accumulating in fp32.  Run here only (CPU).
"""
import sys
import numpy as np
import torch

torch.manual_seed(0)


def make_weights(N, seed=0):
    g = torch.Generator().manual_seed(seed)
    HW = N * N

    def rn(*shape, scale):
        return torch.randn(*shape, generator=g, dtype=torch.float64) * scale

    he = lambda fi: 2.0 / np.sqrt(fi)
    xav = lambda fi, fo: 2.0 / np.sqrt(fi + fo)
    w = {}
    w["conv_w"] = rn(3, 128, scale=he(3))
    w["conv_b"] = torch.zeros(128, dtype=torch.float64)
    for i in range(3):
        w[f"r{i}_w0"] = rn(128, 32, scale=he(128))
        w[f"r{i}_b0"] = torch.zeros(32, dtype=torch.float64)
        w[f"r{i}_dw"] = rn(3, 3, 32, scale=he(9 * 32))
        w[f"r{i}_pw"] = rn(32, 32, scale=he(32))
        w[f"r{i}_b1"] = torch.zeros(32, dtype=torch.float64)
        w[f"r{i}_w2"] = rn(32, 128, scale=he(32))
        w[f"r{i}_b2"] = torch.zeros(128, dtype=torch.float64)
    w["fc0_w"] = rn(128 * HW, 512, scale=he(128 * HW))
    w["fc0_b"] = torch.zeros(512, dtype=torch.float64)
    w["fc1_w"] = rn(512, 512, scale=he(512))
    w["fc1_b"] = torch.zeros(512, dtype=torch.float64)
    w["v_w"] = rn(512, 1, scale=xav(512, 1))
    w["v_b"] = torch.zeros(1, dtype=torch.float64)
    w["p_w"] = rn(512, HW, scale=xav(512, HW))
    w["p_b"] = torch.zeros(HW, dtype=torch.float64)
    return w


def q(x, mode):
    if mode == "f64":
        return x
    if mode == "f32":
        return x.float().double()
    if mode == "f16":
        return x.float().half().double()
    if mode == "bf16":
        return x.float().bfloat16().double()
    if mode == "f16x2":  # hi+lo split, ~22 bits
        hi = x.float().half().double()
        lo = (x - hi).float().half().double()
        return hi + lo
    raise ValueError(mode)


def lrelu(x):
    return torch.where(x > 0, x, 0.2 * x)


def forward(w, inp, N, mode):
    """inp: [B, N, N, 3] float64.  mode: rounding applied to matmul inputs."""
    B = inp.shape[0]
    HW = N * N
    acc = (lambda t: t) if mode == "f64" else (lambda t: t.float().double())
    x = inp.reshape(B, HW, 3)
    x = lrelu(acc(q(x, mode) @ q(w["conv_w"], mode)) + w["conv_b"])
    for i in range(3):
        h = lrelu(acc(q(x, mode) @ q(w[f"r{i}_w0"], mode)) + w[f"r{i}_b0"])
        hp = torch.zeros(B, N + 2, N + 2, 32, dtype=torch.float64)
        hp[:, 1:-1, 1:-1] = q(h, mode).reshape(B, N, N, 32)
        d = torch.zeros(B, N, N, 32, dtype=torch.float64)
        dw = q(w[f"r{i}_dw"], mode)
        for dy in range(3):
            for dx in range(3):
                d += hp[:, dy:dy + N, dx:dx + N] * dw[dy, dx]
        d = acc(d).reshape(B, HW, 32)
        g = lrelu(acc(q(d, mode) @ q(w[f"r{i}_pw"], mode)) + w[f"r{i}_b1"])
        x = lrelu(acc(q(g, mode) @ q(w[f"r{i}_w2"], mode)) + w[f"r{i}_b2"] + x)
    f = x.reshape(B, HW * 128)
    h0 = lrelu(acc(q(f, mode) @ q(w["fc0_w"], mode)) + w["fc0_b"])
    h1 = lrelu(acc(q(h0, mode) @ q(w["fc1_w"], mode)) + w["fc1_b"])
    v = torch.tanh(acc(q(h1, mode) @ q(w["v_w"], mode)) + w["v_b"])
    logits = acc(q(h1, mode) @ q(w["p_w"], mode)) + w["p_b"]
    p = torch.softmax(logits, dim=1)
    return p, v, logits, (x, h0, h1)


def random_inputs(N, B, rng):
    """Positions at random ply depth, encoded like encoder.rs:10-46 (Player mode)."""
    HW = N * N
    out = np.zeros((B, 3 * HW), dtype=np.float64)
    for b in range(B):
        nst = rng.integers(0, HW - 1)
        cells = rng.permutation(HW)[:nst]
        turn_black = (nst % 2 == 0)
        for k, c in enumerate(cells):
            black = (k % 2 == 0)
            mine = (black == turn_black)
            out[b, 2 * c + (0 if mine else 1)] = 1.0
        out[b, 2 * HW:] = 1.0 if turn_black else 0.0
    return torch.from_numpy(out.reshape(B, N, N, 3))


if __name__ == "__main__":
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 15
    B = 48
    rng = np.random.default_rng(1)
    for seed in (0, 1):
        w = make_weights(N, seed)
        inp = random_inputs(N, B, rng)
        p64, v64, l64, acts = forward(w, inp, N, "f64")
        print(f"N={N} seed={seed}: |x|rms={acts[0].pow(2).mean().sqrt():.3f} |h0|rms={acts[1].pow(2).mean().sqrt():.3f} "
              f"|h1|rms={acts[2].pow(2).mean().sqrt():.3f} logits std={l64.std():.3f} pmax={p64.max():.4f} |v|mean={v64.abs().mean():.3f}")
        for mode in ("f32", "f16x2", "f16", "bf16"):
            p, v, l, _ = forward(w, inp, N, mode)
            print(f"   {mode:6s} max|dp|={float((p - p64).abs().max()):.3e}  max|dv|={float((v - v64).abs().max()):.3e}  max|dlogit|={float((l - l64).abs().max()):.3e}")
