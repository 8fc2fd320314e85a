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
