"""Slots mode (omok_selfplay_run_slots: finished slots restart with the next game index, VERDICT r1 "continuous refill keyed by global
game id"): every game's transitions and result must be those of the same game index in an episode of all the games -- compared bit
for bit where the net is row independent (board 9; OMOK_NET_F16X3_ROWS at board 15)."""
import numpy as np
import pytest
import torch

import omok_ai_amd as oa
from omok_ai_amd import binding as B

pytestmark = pytest.mark.gpu


def _episode_records(n, total, sims, k, threshold, mode, seed, max_nodes):
    eng = oa.Engine(board_size=n, games=total, max_nodes=max_nodes, max_tables=max_nodes // 2, max_batch_k=k, seed=seed, net_mode=mode)
    eng.load_random_weights(0)
    sp = oa.SelfPlay(eng)
    sp.reset()
    stats = sp.run(sims, k, 0.25, 0.03, 1.0, threshold)
    rec = sp.replay_record_bytes()
    _, _, plies = sp.game_info()
    _, status, _ = sp.game_info()
    lens = [len(sp.replay(g)[1]) for g in range(total)]
    buf = torch.zeros((sum(lens) + 1) * rec, dtype=torch.uint8, device="cuda")
    assert sp.replay_pack_into(buf.data_ptr(), sum(lens) + 1) == sum(lens)
    out = buf.cpu().numpy().reshape(-1, rec)
    offs = np.concatenate([[0], np.cumsum(lens)])
    games = [out[offs[g]:offs[g + 1]].copy() for g in range(total)]
    eng.close()
    return games, [int(s) for s in status], stats


@pytest.mark.parametrize("n,slots,total,sims,k,mode", [
    (9, 6, 20, 32, 8, B.NET_F16X3),
    (9, 5, 5, 24, 8, B.NET_F16X3),          # total == slots: an ordinary episode through the slots entry point
    (15, 4, 10, 32, 16, B.NET_F16X3_ROWS),
])
def test_slots_mode_reproduces_the_episode_game_by_game(n, slots, total, sims, k, mode):
    seed, threshold, max_nodes = 21, 6, 1024
    want, want_status, estats = _episode_records(n, total, sims, k, threshold, mode, seed, max_nodes)
    eng = oa.Engine(board_size=n, games=slots, max_nodes=max_nodes, max_tables=max_nodes // 2, max_batch_k=k, seed=seed, net_mode=mode)
    eng.load_random_weights(0)
    sp = oa.SelfPlay(eng)
    sp.reset()
    rec = sp.replay_record_bytes()
    cap = total * n * n
    buf = torch.zeros(cap * rec, dtype=torch.uint8, device="cuda")
    stats, nrec, off, ln, status = sp.run_slots(total, sims, k, buf.data_ptr(), cap, 0.25, 0.03, 1.0, threshold)
    out = buf.cpu().numpy().reshape(-1, rec)
    assert nrec == sum(len(w) for w in want)
    assert sp.alive_count == 0
    covered = np.zeros(nrec, dtype=bool)
    for g in range(total):
        assert ln[g] == len(want[g]), f"game {g}: {ln[g]} vs {len(want[g])} transitions"
        assert status[g] == want_status[g]
        got = out[off[g]:off[g] + ln[g]]
        assert np.array_equal(got, want[g]), f"game {g}: records differ"
        assert not covered[off[g]:off[g] + ln[g]].any()
        covered[off[g]:off[g] + ln[g]] = True
    assert covered.all()
    assert stats["sims"] == estats["sims"]     # the same simulations, scheduled differently
    if total > slots:
        assert stats["ply_games"] == estats["ply_games"]
    eng.close()


def test_slots_mode_argument_checks():
    eng = oa.Engine(board_size=9, games=4, max_nodes=256, max_tables=128, max_batch_k=8)
    eng.load_random_weights(0)
    sp = oa.SelfPlay(eng)
    sp.reset()
    buf = torch.zeros(1024 * sp.replay_record_bytes(), dtype=torch.uint8, device="cuda")
    with pytest.raises(B.OmokError):
        sp.run_slots(3, 16, 8, buf.data_ptr(), 1024)      # fewer games than slots
    with pytest.raises(B.OmokError):
        sp.run_slots(8, 16, 8, 0, 1024)                   # no buffer
    sp.run(16, 8, max_plies=1)
    with pytest.raises(B.OmokError):
        sp.run_slots(8, 16, 8, buf.data_ptr(), 1024)      # not a fresh reset
    eng.close()
