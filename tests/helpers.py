"""Shared helpers of the parity tests (no GPU, no product code)."""
import numpy as np


def draw_sequence(n):
    """A legal move sequence that fills the whole n x n board (n in {9, 15}) without ever making an exact five:
    cell (x, y) is Black iff ((x + 2y + off) mod 4) >= 2 -- every row, column and diagonal has runs of at most 2 / 4 -- and
    Black owns one cell more than White, so alternating moves use the cells up exactly.  The last move ends the game as
    GameStatus::Draw (environment/src/lib.rs:160-161)."""
    off = {9: 2, 15: 1}[n]
    black = [y * n + x for y in range(n) for x in range(n) if ((x + 2 * y + off) % 4) >= 2]
    white = [y * n + x for y in range(n) for x in range(n) if ((x + 2 * y + off) % 4) < 2]
    assert len(black) == len(white) + 1
    seq = []
    for i in range(len(white)):
        seq += [black[i], white[i]]
    seq.append(black[-1])
    return seq


def tree_shape(ints):
    """(fully expanded nodes, fully expanded non-root nodes, max depth) of a canonical tree dump
    (ints [n][8] = parent, action, status, turn, legal, nch, n, order|has_policy<<16)."""
    parent, legal, nch = ints[:, 0], ints[:, 4], ints[:, 5]
    full = (nch == legal) & (nch > 0)
    depth = np.zeros(len(ints), dtype=np.int64)
    for i in range(1, len(ints)):
        depth[i] = depth[parent[i]] + 1  # creation order: parent index < child index
    return int(full.sum()), int(full[1:].sum()), int(depth.max()) if len(ints) else 0


def random_positions(n, count, seed):
    """`count` positions in the encoder.rs input layout [count][3 n n]: random stones (0 .. n*n - 2 of them) placed through the
    oracle's rules, random side to move."""
    from oracle import oracle as O
    rng = np.random.default_rng(seed)
    out = np.zeros((count, 3 * n * n), dtype=np.float32)
    for i in range(count):
        env = O.Environment(n)
        for c in rng.permutation(n * n)[: int(rng.integers(0, n * n - 1))]:
            env.place_stone(int(c))
        out[i] = env.encode_nn_input(int(rng.integers(0, 2)))
    return out


def trained_tensors(n, seed, steps=200):
    """Weights after `steps` TrainPhase steps (Adadelta on the augmented replay records of a short self-play episode of the
    random-init net `seed`), on the GPU.  Returns (trained tensors, initial tensors)."""
    import torch
    import omok_ai_amd as oa
    from omok_ai_amd import train as T
    tensors = oa.weights.init_random(n, seed=seed)
    eng = oa.Engine(board_size=n, games=32, max_nodes=512, max_tables=256, max_batch_k=8, seed=4 + seed)
    eng.load_weights(tensors)
    sp = oa.SelfPlay(eng)
    sp.reset()
    sp.run(32, 8)
    _, _, plies = sp.game_info()
    rec = sp.replay_record_bytes()
    total = 6 * int(plies.sum())
    buf = torch.empty(total * rec, dtype=torch.uint8, device="cuda")
    assert sp.replay_augment_into(buf.data_ptr(), total) == total
    ph = T.TrainPhase(n, tensors, "cuda")
    ph.run(buf, update_count=steps, batch_size=128, seed=seed)
    trained = ph.net.tensors()
    eng.close()
    return trained, tensors
