"""Replay post-processing of Trainer::train (src/trainer.rs:207-324): z back-fill + five augmentations per transition.

Pinned by the reference's own 5 symmetry tests (src/utils.rs:70-108, tests/test_oracle_env.py) for the transforms; the
back-fill and the record order have no reference test (parity unpinned there) and follow the cited lines."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import oracle as O  # noqa: E402


def test_oracle_backfill_and_order():
    n, hw, ln = 9, 81, 5
    rng = np.random.default_rng(1)
    boards = rng.integers(0, 3, (ln, hw)).astype(np.uint8)
    turns = (np.arange(ln) % 2).astype(np.uint8)
    pi = rng.random((ln, hw)).astype(np.float32)
    for z_last in (1.0, 0.0):
        z = np.zeros(ln, np.float32)
        z[-1] = z_last
        bo, to, po, zo = O.replay_postprocess(n, boards, turns, pi, z)
        assert bo.shape == (6 * ln, hw)
        # trainer.rs:209-214: last keeps its z, the sign alternates backwards (also for the sign of zero)
        want = np.array([z_last * (-1.0) ** (ln - 1 - t) for t in range(ln)], np.float32)
        assert np.array_equal(zo[:ln].view(np.uint32), want.view(np.uint32))
        assert np.array_equal(bo[:ln], boards) and np.array_equal(po[:ln], pi) and np.array_equal(to[:ln], turns)  # :320
        for t in range(ln):  # :222-318, :321: transition-major, rot90, rot180, rot270, flipH, flipV
            b, p = boards[t].reshape(n, n), pi[t].reshape(n, n)
            for k, f in enumerate((lambda a: np.rot90(a, -1), lambda a: np.rot90(a, 2), lambda a: np.rot90(a, 1),
                                   lambda a: a[:, ::-1], lambda a: a[::-1, :])):
                o = ln + 5 * t + k
                assert np.array_equal(bo[o].reshape(n, n), f(b)) and np.array_equal(po[o].reshape(n, n), f(p))
                assert to[o] == turns[t] and zo[o].tobytes() == zo[t].tobytes()


@pytest.mark.gpu
@pytest.mark.parametrize("n", [9, 15])
def test_device_postprocess_matches_oracle(n):
    import torch
    import omok_ai_amd as oa
    games = 6
    eng = oa.Engine(board_size=n, games=games, max_nodes=512, max_tables=256, max_batch_k=8, seed=11)
    eng.load_random_weights(0)
    sp = oa.SelfPlay(eng)
    sp.reset()
    sp.run(16, 8, 0.25, 0.03, 1.0, 30, 0)  # whole (short-search) episode
    alive, status, plies = sp.game_info()
    assert not alive.any()
    rec = sp.replay_record_bytes()
    total_want = 6 * int(plies.sum())
    buf = torch.zeros((total_want + 7) * rec, dtype=torch.uint8, device="cuda:0")
    total = sp.replay_augment_into(buf.data_ptr(), total_want + 7)
    assert total == total_want
    host = buf.cpu().numpy().reshape(-1, rec)
    hw, brd = n * n, (n * n + 1 + 3) // 4 * 4
    base = 0
    for g in range(games):
        boards, turns, pi, z = sp.replay(g)                      # raw transitions as recorded at play time
        bo, to, po, zo = O.replay_postprocess(n, boards, turns, pi, z)
        gb, gt, gp, gz = sp.replay_augmented(g)                  # per-game accessor
        assert len(gt) == 6 * len(turns)
        assert np.array_equal(gb, bo) and np.array_equal(gt, to)
        assert np.array_equal(gp.view(np.uint32), po.view(np.uint32)) and np.array_equal(gz.view(np.uint32), zo.view(np.uint32))
        blk = host[base:base + len(gt)]                           # packed form, game-id order
        assert np.array_equal(blk[:, :hw], bo) and np.array_equal(blk[:, hw], to)
        assert np.array_equal(blk[:, brd:brd + 4 * hw].copy().view(np.float32).view(np.uint32), po.view(np.uint32))
        assert np.array_equal(blk[:, brd + 4 * hw:brd + 4 * hw + 4].copy().view(np.uint32).ravel(), zo.view(np.uint32))
        base += len(gt)
    assert base == total
    small = torch.zeros(10 * rec, dtype=torch.uint8, device="cuda:0")  # capacity smaller than the output: truncated, no overrun
    assert sp.replay_augment_into(small.data_ptr(), 9) == total_want
    assert not small.cpu().numpy()[9 * rec:].any()
    eng.close()
