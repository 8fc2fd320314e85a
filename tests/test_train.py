"""Training step: alpha-zero/src/agent_model.rs:24-103,136-168, network.rs:249-253, trainer.rs:329-357.

The reference has no test of this path and its arithmetic lives in libtensorflow / tensorflow-rust (absent): parity is
unpinned by the reference.  Chain of checks: oracle/train.py's float64 forward == oracle/net.c (pinned by the golden
vectors); the product's torch graph (run here in float64 on the CPU, test-only) == oracle/train.py for outputs and losses;
autograd gradients == central differences of the oracle's loss; torch's Adadelta == the restated ApplyAdadelta; on the
GPU the float32 step == the float64 step and the updated net, handed to the engine, evaluates like the torch graph."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import omok_ai_amd as oa  # noqa: E402
from omok_ai_amd import train as T  # noqa: E402
from oracle import oracle as O  # noqa: E402
from oracle import train as OT  # noqa: E402


def _batch(n, b, seed):
    rng = np.random.default_rng(seed)
    hw = n * n
    x = np.zeros((b, 3 * hw), np.float32)
    for i in range(b):
        env = O.Environment(n)
        for c in rng.permutation(hw)[: int(rng.integers(0, hw - 1))]:
            env.place_stone(int(c))
        x[i] = env.encode_nn_input(0)
    pi = rng.random((b, hw))
    pi = (pi / pi.sum(axis=1, keepdims=True)).astype(np.float32)
    z = rng.choice([-1.0, 0.0, 1.0], size=(b, 1)).astype(np.float32)
    return x.reshape(b, n, n, 3), pi, z


def test_oracle_forward_matches_the_pinned_c_oracle():
    n = 9
    tensors = oa.weights.init_random(n, seed=3)
    x, _, _ = _batch(n, 6, 0)
    p64, v64 = OT.forward(n, tensors, x)
    pc, vc = O.Net(n, tensors).forward(x.reshape(6, -1), threads=2)
    assert np.abs(p64 - pc.reshape(6, -1)).max() < 1e-4 and np.abs(v64.ravel() - vc.ravel()).max() < 1e-4


def test_torch_graph_losses_and_gradients_match_the_oracle():
    n = 9
    tensors = oa.weights.init_random(n, seed=1)
    tensors = [np.asarray(t, np.float64) * 0.25 for t in tensors]  # keep the softmax away from saturation: gradients of every layer matter
    brng = np.random.default_rng(9)  # zero-initialised biases put empty cells exactly on the LeakyReLU kink, where a central
    tensors = [t + 0.1 * brng.standard_normal(t.shape) if t.ndim == 1 else t for t in tensors]  # difference is not the derivative
    x, pi, z = _batch(n, 5, 1)
    net = T.Network(n, tensors, "cpu", dtype=torch.float64, allow_cpu=True)
    tx, tpi, tz = (torch.as_tensor(a, dtype=torch.float64) for a in (x, pi, z))
    p, v = net(tx)
    p64, v64 = OT.forward(n, tensors, x)
    assert np.abs(p.detach().numpy() - p64).max() < 1e-12 and np.abs(v.detach().numpy() - v64).max() < 1e-12
    pl, vl, ls = net.losses(tx, tpi, tz)
    opl, ovl, ols = OT.losses(n, tensors, x, pi, z)
    assert abs(pl.item() - opl) < 1e-12 and abs(vl.item() - ovl) < 1e-12 and abs(ls.item() - ols) < 1e-12
    ls.backward()
    rng = np.random.default_rng(2)
    for ti in (0, 1, 4, 5, 8, 15, 23, 24, 25, 27, 28, 29, 30):  # conv, depthwise, pointwise, fc and bias variables
        g = net.vars[ti].grad.numpy().ravel()
        for k in rng.integers(0, g.size, size=3):
            h = 1e-6
            tp = [np.array(t, np.float64, copy=True) for t in tensors]
            tm = [np.array(t, np.float64, copy=True) for t in tensors]
            tp[ti].ravel()[k] += h
            tm[ti].ravel()[k] -= h
            fd = (OT.losses(n, tp, x, pi, z)[2] - OT.losses(n, tm, x, pi, z)[2]) / (2 * h)
            assert abs(fd - g[k]) < 1e-7 + 1e-5 * abs(fd), (ti, k, fd, g[k])


def test_adadelta_matches_the_restated_apply_adadelta():
    n = 9
    tensors = [np.asarray(t, np.float64) * 0.25 for t in oa.weights.init_random(n, seed=5)]
    x, pi, z = _batch(n, 4, 3)
    ph = T.TrainPhase(n, tensors, "cpu", dtype=torch.float64, allow_cpu=True)
    var = [np.array(t, np.float64).ravel() for t in tensors]
    acc = [np.zeros_like(a) for a in var]
    accu = [np.zeros_like(a) for a in var]
    tx, tpi, tz = (torch.as_tensor(a, dtype=torch.float64) for a in (x, pi, z))
    for _ in range(3):
        ph.opt.zero_grad()
        ph.net.losses(tx, tpi, tz)[2].backward()
        grads = [p.grad.numpy().ravel().copy() for p in ph.net.vars]
        losses = ph.step(tx, tpi, tz)                                   # minimize, then the losses after the update
        for i in range(31):
            var[i], acc[i], accu[i] = OT.adadelta_apply(var[i], acc[i], accu[i], grads[i], T.LEARNING_RATE, T.RHO, T.EPSILON)
            assert np.abs(ph.net.vars[i].detach().numpy().ravel() - var[i]).max() < 1e-13
        want = OT.losses(n, [v.reshape(s) for v, s in zip(var, OT.shapes(n))], x, pi, z)
        assert abs(losses[0] - want[0]) < 1e-10 and abs(losses[1] - want[1]) < 1e-10 and abs(losses[2] - want[2]) < 1e-10


def test_record_decoding_matches_encode_nn_input():
    n = 9
    hw = n * n
    brd = (hw + 1 + 3) // 4 * 4
    rec = brd + 4 * hw + 4
    rng = np.random.default_rng(4)
    r = np.zeros((7, rec), np.uint8)
    want_x, want_pi, want_z = [], [], []
    for i in range(7):
        board = rng.integers(0, 3, hw).astype(np.uint8)
        turn = int(rng.integers(0, 2))
        pi = rng.random(hw).astype(np.float32)
        z = np.float32(rng.choice([-1.0, 0.0, 1.0]))
        r[i, :hw] = board
        r[i, hw] = turn
        r[i, brd:brd + 4 * hw] = pi.view(np.uint8)
        r[i, brd + 4 * hw:] = np.array([z], np.float32).view(np.uint8)
        want_x.append(OT.encode_input(n, board, turn))
        want_pi.append(pi)
        want_z.append(z)
    x, pi, z = T.decode_records(torch.from_numpy(r), n)
    assert np.array_equal(x.numpy().reshape(7, -1), np.stack(want_x))
    assert np.array_equal(pi.numpy(), np.stack(want_pi)) and np.array_equal(z.numpy().ravel(), np.array(want_z, np.float32))


def test_no_cpu_training_path_in_the_product():
    with pytest.raises(RuntimeError):
        T.TrainPhase(9, oa.weights.init_random(9, 0), "cpu")


@pytest.mark.gpu
def test_gpu_training_phase_end_to_end():
    n, games = 9, 24
    eng = oa.Engine(board_size=n, games=games, max_nodes=512, max_tables=256, max_batch_k=8, seed=5)
    tensors = oa.weights.init_random(n, seed=0)
    eng.load_weights(tensors)
    sp = oa.SelfPlay(eng)
    sp.reset()
    sp.run(16, 8, 0.25, 0.03, 1.0, 30, 0)
    _, _, plies = sp.game_info()
    rec = sp.replay_record_bytes()
    total = 6 * int(plies.sum())
    buf = torch.zeros(total * rec, dtype=torch.uint8, device="cuda:0")
    assert sp.replay_augment_into(buf.data_ptr(), total) == total
    # one float32 step on the GPU == the same step in float64 on the CPU (same records, same batch)
    idx = torch.arange(0, min(64, total), device="cuda:0")
    x, pi, z = T.decode_records(buf.reshape(-1, rec)[idx], n)
    gpu = T.TrainPhase(n, tensors, "cuda:0")
    cpu = T.TrainPhase(n, tensors, "cpu", dtype=torch.float64, allow_cpu=True)
    lg = gpu.step(x, pi, z)
    lc = cpu.step(x.cpu().double(), pi.cpu().double(), z.cpu().double())
    assert max(abs(a - b) for a, b in zip(lg, lc)) < 2e-3 * max(1.0, abs(lc[2]))
    for a, b in zip(gpu.net.vars, cpu.net.vars):
        assert np.abs(a.detach().cpu().numpy() - b.detach().numpy()).max() < 5e-4  # |update| <= lr * sqrt(eps-ratio) * ...: small steps
    # a short phase lowers the loss on its own replay memory, and the engine then evaluates like the torch graph
    first = gpu.step(x, pi, z)[2]
    v_loss, p_loss, loss = gpu.run(buf, update_count=30, batch_size=64, seed=1)
    assert np.isfinite(loss) and gpu.step(x, pi, z)[2] < first
    gpu.push_to(eng)
    xe = x[:16].reshape(16, -1).cpu().numpy()
    pe, ve = eng.evaluate_pv(xe)
    with torch.no_grad():
        pt, vt = gpu.net(x[:16])
    assert np.abs(pe.reshape(16, -1) - pt.cpu().numpy()).max() < 1e-3 and np.abs(ve.ravel() - vt.cpu().numpy().ravel()).max() < 1e-3
    eng.close()


@pytest.mark.gpu
def test_trainer_iterations_save_and_resume(tmp_path):
    """Trainer::train mirror (src/trainer.rs:69-386): self-play -> post-process -> train -> save, and a new Trainer resumes
    from the saved ModelIO file."""
    from omok_ai_amd import trainer as TR
    from oracle import model_io as M
    p = TR.Parameters(model_name="tiny", episode_count=8, evaluate_count=16, evaluate_batch_size=8,
                      parameter_update_count=5, parameter_update_batch_size=32, replay_memory_size=500)
    save_dir = str(tmp_path / "saves")
    tr = TR.Trainer(p, board_size=9, seed=3, save_dir=save_dir)
    w0 = tr.phase.net.tensors()
    logs = []
    v_loss, p_loss, loss = tr.train(2, log=logs.append)
    assert len(logs) == 2 and np.isfinite(loss) and abs(loss - (v_loss + p_loss)) < 1e-4 * max(1.0, abs(loss))
    w2 = tr.phase.net.tensors()
    assert any(not np.array_equal(a, b) for a, b in zip(w0, w2))          # the variables moved
    names, params = M.model_load(os.path.join(save_dir, "tiny"))          # Trainer::save wrote the reference's format
    assert len(params) == 31
    for a, b in zip(w2, params):
        assert np.array_equal(np.asarray(a, np.float32).ravel().view(np.uint32), b.view(np.uint32))
    tr.close()
    tr2 = TR.Trainer(p, board_size=9, seed=99, save_dir=save_dir)           # Trainer::new -> load(model_name)
    for a, b in zip(tr2.phase.net.tensors(), w2):
        assert np.array_equal(np.asarray(a, np.float32).view(np.uint32), np.asarray(b, np.float32).view(np.uint32))
    tr2.close()
