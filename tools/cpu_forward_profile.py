"""Where the CPU restatement's net time goes on a host (bench.py's cpu_baseline leg uses the torch-CPU graph of omok-ai_amd/train.py): times the trunk (conv2d on
NCHW), fc0 and the tail of a 4096-row forward at several thread counts, and an NHWC matmul form of the 1x1 convolutions.  usage: python tools/cpu_forward_profile.py [rows]"""
import os, sys, time
import numpy as np, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import omok_ai_amd as oa
n = 15; hw = n * n
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
tensors = oa.weights.init_random(n, seed=0)
t = [torch.from_numpy(np.asarray(a, dtype=np.float32).reshape(s)) for a, s in zip(tensors, oa.weights.tensor_shapes(n))]
x = torch.from_numpy((np.random.default_rng(0).random((B, n, n, 3)) < 0.1).astype(np.float32))
lr = lambda z: F.leaky_relu(z, 0.2)
def trunk_nchw(x):
    a = x.permute(0, 3, 1, 2)
    conv = lambda a, w, b: F.conv2d(a, w.permute(3, 2, 0, 1), b)
    a = lr(conv(a, t[0], t[1]))
    for i in range(3):
        w0, b0, dw, pw, b1, w2, b2 = t[2 + 7 * i: 9 + 7 * i]
        h = lr(conv(a, w0, b0)); d = F.conv2d(h, dw.permute(2, 3, 0, 1), None, padding=1, groups=dw.shape[2])
        a = lr(conv(lr(conv(d, pw, b1)), w2, b2) + a)
    return a.permute(0, 2, 3, 1).reshape(a.shape[0], -1)
def trunk_mm(x):  # 1x1 convolutions as matmuls on [B*HW, C]; depthwise still conv2d
    a = lr(x.reshape(-1, 3) @ t[0].reshape(3, -1) + t[1])
    for i in range(3):
        w0, b0, dw, pw, b1, w2, b2 = t[2 + 7 * i: 9 + 7 * i]
        h = lr(a @ w0.reshape(w0.shape[2], -1) + b0)
        d = F.conv2d(h.reshape(-1, n, n, h.shape[1]).permute(0, 3, 1, 2), dw.permute(2, 3, 0, 1), None, padding=1, groups=dw.shape[2]).permute(0, 2, 3, 1).reshape(-1, h.shape[1])
        a = lr(lr(d @ pw.reshape(pw.shape[2], -1) + b1) @ w2.reshape(w2.shape[2], -1) + b2 + a)
    return a.reshape(-1, hw * a.shape[1])
def tm(f, *a):
    f(*a); t0 = time.perf_counter(); r = f(*a); return time.perf_counter() - t0, r
print(f"host threads {os.cpu_count()}, rows {B}")
with torch.no_grad():
    for th in (8, 16, 32, 64, 128, os.cpu_count()):
        if th > (os.cpu_count() or 1): continue
        torch.set_num_threads(th)
        t1, f1 = tm(trunk_nchw, x); t2, f2 = tm(trunk_mm, x)
        t3, h0 = tm(lambda f: lr(f @ t[23] + t[24]), f1)
        t4, _ = tm(lambda h: (lr(h @ t[25] + t[26]) @ t[29]), h0)
        print(f"threads {th:4d}: trunk conv2d {t1*1e3:8.1f} ms  trunk matmul-form {t2*1e3:8.1f} ms  fc0 {t3*1e3:8.1f} ms ({2*B*128*hw*512/t3/1e12:.2f} TFLOP/s)  fc1+policy head {t4*1e3:7.1f} ms  "
              f"-> {B/(min(t1,t2)+t3+t4):9.0f} rows/s   (|trunk forms differ| {float((f1-f2).abs().max()):.1e})", flush=True)
