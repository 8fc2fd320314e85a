"""CPU restatement of the training-step arithmetic (TEST INFRASTRUCTURE ONLY; see oracle/oracle.py for who may import).

numpy float64, loops only where numpy has no primitive.  Follows
  alpha-zero/src/network.rs:51-262 (graph; same op order as oracle/net.c, which the golden vectors pin),
  network.rs:249-253 (p_loss = mean_b softmax_cross_entropy_with_logits(logits, labels = pi)),
  alpha-zero/src/agent_model.rs:57-73 (v_loss = mean((z - v)^2), loss = v_loss + p_loss),
  agent_model.rs:24,75-82 (AdadeltaOptimizer, lr 0.01).
The optimizer and every op kernel live in third-party code absent from the reference tree (tensorflow 0.21.0 /
tensorflow-sys 0.24.0, Cargo.lock): restated from the published semantics -- ApplyAdadelta:
    accum        = rho * accum + (1 - rho) * grad^2
    update       = sqrt(accum_update + eps) / sqrt(accum + eps) * grad
    var         -= lr * update
    accum_update = rho * accum_update + (1 - rho) * update^2
with tensorflow-rust's AdadeltaOptimizer defaults rho = 0.95, eps = 1e-8.  The reference has no test of the training step:
PARITY UNPINNED BY THE REFERENCE.  The forward part is pinned through oracle/net.c (tests/test_train.py compares them);
there is no backward pass here: gradients are checked by central differences of `losses`.
"""
import numpy as np

C, M, FC = 128, 32, 512


def _lrelu(a):
    return np.where(a > 0, a, 0.2 * a)


def shapes(n):
    hw = n * n
    s = [(1, 1, 3, C), (C,)]
    for _ in range(3):
        s += [(1, 1, C, M), (M,), (3, 3, M, 1), (1, 1, M, M), (M,), (1, 1, M, C), (C,)]
    return s + [(C * hw, FC), (FC,), (FC, FC), (FC,), (FC, 1), (1,), (FC, hw), (hw,)]


def logits_v(n, tensors, x):
    """x [B, N, N, 3] -> (logits [B, HW], v [B, 1]); NHWC throughout."""
    t = [np.asarray(a, dtype=np.float64).reshape(s) for a, s in zip(tensors, shapes(n))]
    a = np.asarray(x, dtype=np.float64).reshape(-1, n, n, 3)
    a = _lrelu(a @ t[0][0, 0] + t[1])
    for i in range(3):
        w0, b0, dw, pw, b1, w2, b2 = t[2 + 7 * i: 9 + 7 * i]
        h = _lrelu(a @ w0[0, 0] + b0)
        hp = np.pad(h, ((0, 0), (1, 1), (1, 1), (0, 0)))
        d = np.zeros_like(h)
        for dy in range(3):          # depthwise 3x3, SAME, stride 1, no bias
            for dx in range(3):
                d += hp[:, dy:dy + n, dx:dx + n, :] * dw[dy, dx, :, 0]
        g = _lrelu(d @ pw[0, 0] + b1)
        a = _lrelu((g @ w2[0, 0] + b2) + a)
    f = a.reshape(a.shape[0], -1)
    h0 = _lrelu(f @ t[23] + t[24])
    h1 = _lrelu(h0 @ t[25] + t[26])
    return h1 @ t[29] + t[30], np.tanh(h1 @ t[27] + t[28])


def forward(n, tensors, x):
    lg, v = logits_v(n, tensors, x)
    e = np.exp(lg - lg.max(axis=1, keepdims=True))
    return e / e.sum(axis=1, keepdims=True), v


def losses(n, tensors, x, pi, z):
    lg, v = logits_v(n, tensors, x)
    m = lg.max(axis=1, keepdims=True)
    logp = lg - m - np.log(np.exp(lg - m).sum(axis=1, keepdims=True))
    p_loss = (-(np.asarray(pi, np.float64).reshape(lg.shape) * logp).sum(axis=1)).mean()
    v_loss = ((np.asarray(z, np.float64).reshape(v.shape) - v) ** 2).mean()
    return p_loss, v_loss, v_loss + p_loss


def adadelta_apply(var, accum, accum_update, grad, lr=0.01, rho=0.95, eps=1e-8):
    accum = rho * accum + (1.0 - rho) * grad * grad
    update = np.sqrt(accum_update + eps) / np.sqrt(accum + eps) * grad
    var = var - lr * update
    accum_update = rho * accum_update + (1.0 - rho) * update * update
    return var, accum, accum_update


def encode_input(n, board, turn):
    """encode_nn_input(EnvTurnMode::Player) for one record (encoder.rs:10-46, environment lib.rs:81-102)."""
    hw = n * n
    f = np.zeros(3 * hw, dtype=np.float32)
    own = 1 if turn == 0 else 2
    for i in range(hw):
        if board[i] == own:
            f[2 * i] = 1.0
        elif board[i] != 0:
            f[2 * i + 1] = 1.0
    f[2 * hw:] = 1.0 if turn == 0 else 0.0
    return f
