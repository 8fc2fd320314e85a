"""Generates tests/golden/*.npz (run HERE, on CPU; the fixtures are committed).

1. net_n{9,15}.npz — an INDEPENDENT torch implementation (F.conv2d NCHW / groups= depthwise /
   F.leaky_relu(0.2) / softmax / tanh, float64) of alpha-zero/src/network.rs:51-262 evaluated on
   seeded weights and encoded positions.  It pins the oracle's fp32 restatement (oracle/net.c).
   The reference itself (Rust + libtensorflow) cannot run here, so these are cross-check
   vectors, not reference outputs.
2. selfplay_n9.npz — move sequences / final tree fingerprints of small oracle self-play runs
   (regression pins for oracle/selfplay.c; the reference has no tests for mcts/executor/agent).
3. selfplay_n15.npz — the regime the headline runs in (15x15, 800 simulations per move in 50 rounds of K = 16: fully
   expanded nodes, depth >= 3), generated through the LITERAL restatement (oracle/literal.c: the reference's own data
   structures) and checked against oracle/selfplay.c by tests/test_oracle_selfplay.py: moves and canonical tree dumps
   of the first plies of two games.
"""
import hashlib
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import omok_ai_amd  # noqa: E402,F401
from omok_ai_amd import weights  # noqa: E402
from oracle import oracle  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def torch_forward(tensors, inp, n):
    t = [torch.from_numpy(np.asarray(x, dtype=np.float64)) for x in tensors]
    x = torch.from_numpy(np.asarray(inp, dtype=np.float64)).reshape(-1, n, n, 3).permute(0, 3, 1, 2)

    def conv(x, w, b):  # HWIO -> OIHW
        return F.conv2d(x, w.permute(3, 2, 0, 1), b, padding="same")

    x = F.leaky_relu(conv(x, t[0], t[1]), 0.2)
    for i in range(3):
        w0, b0, dw, pw, b1, w2, b2 = t[2 + 7 * i: 9 + 7 * i]
        h = F.leaky_relu(conv(x, w0, b0), 0.2)
        d = F.conv2d(h, dw.permute(2, 3, 0, 1), None, padding=1, groups=32)  # [3,3,32,1] -> [32,1,3,3]
        g = F.leaky_relu(conv(d, pw, b1), 0.2)
        x = F.leaky_relu(conv(g, w2, b2) + x, 0.2)
    f = x.permute(0, 2, 3, 1).reshape(x.shape[0], -1)  # NHWC flatten
    h0 = F.leaky_relu(f @ t[23] + t[24], 0.2)
    h1 = F.leaky_relu(h0 @ t[25] + t[26], 0.2)
    v = torch.tanh(h1 @ t[27] + t[28]).reshape(-1)
    p = torch.softmax(h1 @ t[29] + t[30], dim=1)
    return p.numpy(), v.numpy()


def random_positions(n, count, rng):
    out = []
    for _ in range(count):
        env = oracle.Environment(n)
        nst = int(rng.integers(0, n * n - 1))
        for c in rng.permutation(n * n)[:nst]:
            env.place_stone(int(c))
        out.append(env.encode_nn_input(int(rng.integers(0, 2))))
    return np.stack(out)


def make_net(n, count):
    tensors = weights.init_random(n, seed=0)
    rng = np.random.default_rng(100 + n)
    inp = random_positions(n, count, rng)
    inp[0] = oracle.Environment(n).encode_nn_input(0)  # empty board (Agent::new)
    p, v = torch_forward(tensors, inp, n)
    np.savez_compressed(os.path.join(OUT, f"net_n{n}.npz"), n=n, seed=0, inputs=inp.astype(np.float32),
                        p=p, v=v, weight_checksum=weights.checksum(tensors))
    print(f"net_n{n}: {count} positions, pmax={p.max():.4f}")


def fingerprint(sp, games):
    h = hashlib.sha256()
    for g in range(games):
        for side in (0, 1):
            ints, floats = sp.tree_dump(g, side)
            h.update(ints.tobytes())
            h.update(floats.tobytes())
            h.update(np.array(sp.tree_root(g, side)[:1], dtype=np.uint32).tobytes())
    return h.hexdigest()


def make_selfplay():
    n, games, count, k = 9, 3, 48, 8
    tensors = weights.init_random(n, seed=0)
    net = oracle.Net(n, tensors)
    root_p, _ = net.forward(oracle.Environment(n).encode_nn_input(0)[None])
    sp = oracle.SelfPlay(n, games, cap_nodes=2048, cap_tables=1024, seed=7)
    sp.reset(root_p[0])
    actions, prints = [], []
    while sp.alive_count > 0:
        for rnd in range((count + k - 1) // k):
            inp = sp.round_generate(rnd, k, 0.25, 0.03)
            if len(inp) == 0:
                continue
            p, v = net.forward(inp)
            sp.round_scatter(p, v)
        actions.append(sp.sample(1.0, 6))
        m = sp.mirror_generate()
        p, _ = net.forward(m)
        sp.advance(p)
        prints.append(fingerprint(sp, games))
    assert sp.error == 0
    np.savez_compressed(os.path.join(OUT, "selfplay_n9.npz"), n=n, games=games, count=count, k=k, seed=7,
                        threshold=6, actions=np.stack(actions), fingerprints=np.array(prints),
                        status=np.array([sp.game_status(g) for g in range(games)]),
                        plies=np.array([sp.game_plies(g) for g in range(games)]))
    print("selfplay_n9: plies", [sp.game_plies(g) for g in range(games)], "status", [sp.game_status(g) for g in range(games)])


def fingerprint_dumps(sp, games):
    """sha256 over the canonical dumps of both trees of every game (the surface oracle.SelfPlay and oracle.Literal share)"""
    h = hashlib.sha256()
    for g in range(games):
        for side in (0, 1):
            ints, floats = sp.tree_dump(g, side)
            h.update(np.ascontiguousarray(ints[:, :7]).tobytes())  # node records
            h.update(np.ascontiguousarray(ints[:, 7] & 0xFFFF).tobytes())  # insertion ranks (the high half is per-oracle bookkeeping)
            h.update(floats.tobytes())  # w and policy bits
    return h.hexdigest()


def make_selfplay_n15():
    n, games, count, k, plies, seed = 15, 2, 800, 16, 4, 15
    tensors = weights.init_random(n, seed=0)
    net = oracle.Net(n, tensors)
    root_p, _ = net.forward(oracle.Environment(n).encode_nn_input(0)[None])
    sp = oracle.Literal(n, games, seed=seed, cap_nodes=8192)
    sp.reset(root_p[0])
    actions, prints, nodes, depth = [], [], [], []
    for _ in range(plies):
        for rnd in range(count // k):
            inp, _ = sp.round_generate(rnd, k, 0.25, 0.03)
            if len(inp):
                p, v = net.forward(inp, threads=8)
                sp.round_scatter(p, v)
        side = sp.ply & 1
        for g in range(games):  # the searched trees' shape: what makes this the headline's regime
            ints, _ = sp.tree_dump(g, side)
            d = np.zeros(len(ints), dtype=np.int64)
            for i in range(1, len(ints)):
                d[i] = d[ints[i, 0]] + 1
            nodes.append(len(ints))
            depth.append(int(d.max()))
        actions.append(sp.sample(1.0, 30))
        m, _ = sp.mirror_generate()
        p, _ = net.forward(m, threads=8)
        sp.advance(p)
        prints.append(fingerprint_dumps(sp, games))
    assert sp.error == 0
    np.savez_compressed(os.path.join(OUT, "selfplay_n15.npz"), n=n, games=games, count=count, k=k, seed=seed, threshold=30, plies=plies,
                        actions=np.stack(actions), fingerprints=np.array(prints), searched_nodes=np.array(nodes), searched_depth=np.array(depth))
    print("selfplay_n15: nodes of the searched trees", nodes, "depth", depth)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    if "n15" in sys.argv[1:]:  # (only the new fixture: the others are committed and stay as they are)
        make_selfplay_n15()
        sys.exit(0)
    make_net(9, 8)
    make_net(15, 4)
    make_selfplay()
    make_selfplay_n15()
