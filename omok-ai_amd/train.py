"""Training phase of one Trainer iteration on the GPU, PyTorch-ROCm autograd.

Reference (file:line):
  alpha-zero/src/network.rs:51-262        the graph (same 31 variables, same order and shapes as the engine's net)
  alpha-zero/src/network.rs:249-253       p_loss = mean_b softmax_cross_entropy_with_logits(labels = pi, logits)
  alpha-zero/src/agent_model.rs:57-67     v_loss = mean((z - v)^2);  :69-73 loss = v_loss + p_loss
  alpha-zero/src/agent_model.rs:24,75-82  AdadeltaOptimizer, learning rate 0.01, other settings at the optimizer's defaults
                                          (tensorflow 0.21.0 train.rs: rho 0.95, epsilon 1e-8; ApplyAdadelta semantics)
  alpha-zero/src/agent_model.rs:136-168   AgentModel::train: one minimize run, THEN the three losses are evaluated (after the update)
  src/trainer.rs:329-357                  parameter_update_count steps, each on parameter_update_batch_size transitions drawn
                                          without replacement from the replay memory (choose_multiple), encoded with
                                          encode_nn_input(Player) / encode_nn_targets (encoder.rs:10-68)
The self-play engine is the HIP library; this module only needs torch on the same GPU and hands the updated tensors back
through omok_net_load/commit.  Data-parallel over GPUs: gradients are averaged with one RCCL all-reduce per step.
There is no CPU training path in the product: `device` must be a CUDA/HIP device unless `allow_cpu` (tests) is set.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import weights as W

LEARNING_RATE = 0.01   # AgentModel::LEARNING_RATE, agent_model.rs:24
RHO = 0.95             # tensorflow-rust AdadeltaOptimizer default
EPSILON = 1e-8         # tensorflow-rust AdadeltaOptimizer default


class Network(torch.nn.Module):
    """The 31 variables in the reference's order and shapes (conv kernels HWIO, fc [in, out])."""

    def __init__(self, n, tensors, device, dtype=torch.float32, allow_cpu=False):
        super().__init__()
        device = torch.device(device)
        if device.type != "cuda" and not allow_cpu:
            raise RuntimeError("omok_ai_amd.train runs on the GPU only (no CPU training path)")
        self.n = n
        shapes = W.tensor_shapes(n)
        assert len(tensors) == len(shapes) == 31
        self.vars = torch.nn.ParameterList(
            [torch.nn.Parameter(torch.as_tensor(np.asarray(t, dtype=np.float64).reshape(s), dtype=dtype, device=device))
             for t, s in zip(tensors, shapes)])

    def logits_v(self, x):
        """x [B, N, N, 3] (encoder.rs layout) -> (policy logits [B, N*N], v [B, 1])."""
        t = list(self.vars)
        x = x.permute(0, 3, 1, 2)

        def conv(a, w, b):  # 1x1 conv2d + BiasAdd, NHWC weights HWIO (network-utils/src/lib.rs:95-170)
            return F.conv2d(a, w.permute(3, 2, 0, 1), b)

        x = F.leaky_relu(conv(x, t[0], t[1]), 0.2)
        for i in range(3):  # bottleneck residual blocks (network-utils/src/lib.rs:386-461)
            w0, b0, dw, pw, b1, w2, b2 = t[2 + 7 * i: 9 + 7 * i]
            h = F.leaky_relu(conv(x, w0, b0), 0.2)
            d = F.conv2d(h, dw.permute(2, 3, 0, 1), None, padding=1, groups=dw.shape[2])  # depthwise 3x3 SAME (:172-262)
            g = F.leaky_relu(conv(d, pw, b1), 0.2)
            x = F.leaky_relu(conv(g, w2, b2) + x, 0.2)  # add before the activation (network.rs:108-111)
        f = x.permute(0, 2, 3, 1).reshape(x.shape[0], -1)  # NHWC flatten
        h0 = F.leaky_relu(f @ t[23] + t[24], 0.2)
        h1 = F.leaky_relu(h0 @ t[25] + t[26], 0.2)
        v = torch.tanh(h1 @ t[27] + t[28])
        return h1 @ t[29] + t[30], v

    def forward(self, x):
        logits, v = self.logits_v(x)
        return torch.softmax(logits, dim=1), v

    def losses(self, x, pi, z):
        """(p_loss, v_loss, loss) of network.rs:249-253 / agent_model.rs:57-73; pi [B, N*N], z [B, 1]."""
        logits, v = self.logits_v(x)
        p_loss = (-(pi * F.log_softmax(logits, dim=1)).sum(dim=1)).mean()
        v_loss = ((z - v) ** 2).mean()
        return p_loss, v_loss, v_loss + p_loss

    def tensors(self):
        return [p.detach().to(torch.float32).cpu().numpy() for p in self.vars]


def decode_records(records, n):
    """Packed replay records (omok_replay_pack_dev / omok_replay_augment_dev layout: board u8[HW], turn u8, pad to 4, pi
    f32[HW], z f32) as a uint8 tensor [R, REC] -> (input [R,N,N,3], pi [R,HW], z [R,1]) on the records' device.
    Input = encode_nn_input(EnvTurnMode::Player) (encoder.rs:10-46, environment lib.rs:81-102): plane pair per cell =
    (stone of the side to move, stone of the opponent), third plane 1 where Black is to move."""
    hw = n * n
    brd = (hw + 1 + 3) // 4 * 4
    rec = brd + 4 * hw + 4
    records = records.reshape(-1, rec)
    board = records[:, :hw]
    turn = records[:, hw].to(torch.int64)                      # Turn::Black = 0, Turn::White = 1
    mine = (board == (turn + 1).unsqueeze(1).to(torch.uint8))  # Stone::Black = 1, Stone::White = 2
    theirs = (board != 0) & ~mine
    black_to_move = (turn == 0).to(torch.float32).unsqueeze(1).expand(-1, hw)
    x = torch.empty((records.shape[0], 3 * hw), dtype=torch.float32, device=records.device)
    x[:, 0:2 * hw:2] = mine.to(torch.float32)
    x[:, 1:2 * hw:2] = theirs.to(torch.float32)
    x[:, 2 * hw:] = black_to_move
    pi = records[:, brd:brd + 4 * hw].contiguous().view(torch.float32)
    z = records[:, brd + 4 * hw:brd + 4 * hw + 4].contiguous().view(torch.float32)
    return x.reshape(-1, n, n, 3), pi, z


class TrainPhase:
    """AgentModel::train over a replay memory held on the device as packed records."""

    def __init__(self, n, tensors, device, dtype=torch.float32, allow_cpu=False):
        self.n = n
        self.net = Network(n, tensors, device, dtype=dtype, allow_cpu=allow_cpu)
        # one accumulator pair per variable, zero-initialised (AdadeltaOptimizer::minimize creates them per variable)
        self.opt = torch.optim.Adadelta(self.net.parameters(), lr=LEARNING_RATE, rho=RHO, eps=EPSILON, weight_decay=0.0)

    def step(self, x, pi, z):
        """One AgentModel::train call (agent_model.rs:136-168): minimize, then the losses evaluated after the update."""
        self.opt.zero_grad(set_to_none=True)
        _, _, loss = self.net.losses(x, pi, z)
        loss.backward()
        if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
            flat = torch.cat([p.grad.reshape(-1) for p in self.net.parameters()])  # data-parallel: one all-reduce per step
            torch.distributed.all_reduce(flat)
            flat /= torch.distributed.get_world_size()
            o = 0
            for p in self.net.parameters():
                p.grad.copy_(flat[o:o + p.numel()].view_as(p))
                o += p.numel()
        self.opt.step()
        with torch.no_grad():
            p_loss, v_loss, loss = self.net.losses(x, pi, z)
        return float(p_loss), float(v_loss), float(loss)

    def run(self, records, update_count=600, batch_size=128, seed=0):
        """trainer.rs:329-357: `update_count` steps on `batch_size` records drawn without replacement; returns the mean
        (v_loss, p_loss, loss) over the last 100 steps like the reference's log line (:354-362)."""
        hw = self.n * self.n
        rec = (hw + 1 + 3) // 4 * 4 + 4 * hw + 4
        records = records.reshape(-1, rec)
        g = torch.Generator(device=records.device)
        g.manual_seed(seed)
        recent = []
        for _ in range(update_count):
            k = min(batch_size, records.shape[0])
            idx = torch.randperm(records.shape[0], generator=g, device=records.device)[:k]
            x, pi, z = decode_records(records[idx], self.n)
            dt = next(self.net.parameters()).dtype
            p_loss, v_loss, loss = self.step(x.to(dt), pi.to(dt), z.to(dt))
            recent.append((v_loss, p_loss, loss))
            recent = recent[-100:]
        a = np.asarray(recent, dtype=np.float64)
        return tuple(a.mean(axis=0)) if len(a) else (0.0, 0.0, 0.0)

    def push_to(self, engine):
        """Hand the updated variables to the self-play engine (omok_net_load x31 + omok_net_commit)."""
        engine.load_weights(self.net.tensors())
