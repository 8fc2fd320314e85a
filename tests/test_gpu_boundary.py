"""GPU tests of the boundary entry points beyond the self-play loop (run with -m gpu): externally chosen moves
(Agent::ensure_action_exists + play_action on both agents), compute_policy, root children, place_stone on caller-held
environments, the per-episode RNG stream, the packed replay buffer, argument / state errors.  Everything through the C ABI,
checked against the oracle (and, for the external-move path, against the literal second oracle as well)."""
import numpy as np
import pytest
import torch

import omok_ai_amd as oa
from omok_ai_amd import binding as B
from oracle import oracle as O
from helpers import draw_sequence

pytestmark = pytest.mark.gpu


def _dumps_equal(sp, osp, games, tag):
    for g in range(games):
        for side in (0, 1):
            gi, gf = sp.tree_dump(g, side)
            oi, of = osp.tree_dump(g, side)
            assert gi.shape == oi.shape and np.array_equal(gi, oi), f"{tag}: node records (game {g} side {side})"
            assert np.array_equal(gf.view(np.uint32), of.view(np.uint32)), f"{tag}: w / policy bits (game {g} side {side})"


def _engine(n, games, k, seed=3, max_nodes=1024, max_tables=512, **kw):
    eng = oa.Engine(board_size=n, games=games, max_nodes=max_nodes, max_tables=max_tables, max_batch_k=k, seed=seed, **kw)
    eng.load_random_weights(0)
    return eng, oa.SelfPlay(eng)


def _search(sp, osp, count, k):
    for rnd in range((count + k - 1) // k):
        nreq = sp.round_generate(rnd, k, 0.25, 0.03)
        oin = osp.round_generate(rnd, k, 0.25, 0.03)
        assert nreq == len(oin) and np.array_equal(sp.round_inputs(), oin)
        p, v = sp.round_eval()
        sp.round_scatter()
        osp.round_scatter(p, v)


def test_external_moves_match_both_oracles():
    """Every other ply the move comes from the caller (a random empty cell, usually not in either tree):
    omok_set_actions + the mirror step == ensure_action_exists + play_action on both agents in the oracles."""
    n, games, count, k = 9, 4, 32, 8
    hw = n * n
    eng, sp = _engine(n, games, k)
    sp.reset()
    root_p = eng.evaluate_p(O.Environment(n).encode_nn_input(0)[None]).reshape(-1)
    osp = O.SelfPlay(n, games, cap_nodes=1024, cap_tables=512, seed=3)
    lit = O.Literal(n, games, seed=3, cap_nodes=1024)
    osp.reset(root_p)
    lit.reset(root_p)
    rng = np.random.default_rng(0)
    played = [set() for _ in range(games)]
    for ply in range(14):
        # search on all three (the literal takes its rows in slot order)
        for rnd in range(count // k):
            nreq = sp.round_generate(rnd, k, 0.25, 0.03)
            oin = osp.round_generate(rnd, k, 0.25, 0.03)
            lin, lg = lit.round_generate(rnd, k, 0.25, 0.03)
            assert nreq == len(oin) == len(lin) and np.array_equal(sp.round_inputs(), oin)
            og = [osp.request_info(r)[0] for r in range(nreq)]
            perm = _perm(og, lg)
            assert np.array_equal(lin, oin[perm])
            p, v = sp.round_eval()
            sp.round_scatter()
            osp.round_scatter(p, v)
            lit.round_scatter(p[perm], v[perm])
        pi, has = sp.compute_policy()
        for g in range(games):
            want = osp.compute_policy(g)
            assert bool(has[g]) == (want is not None)
            if want is not None:
                assert np.array_equal(pi[g].view(np.uint32), want.view(np.uint32))
                a, cn, cw, cp = sp.root_children(g, ply & 1)
                ints, floats = sp.tree_dump(g, ply & 1)
                kids = np.flatnonzero(ints[:, 0] == 0)
                order = np.argsort(ints[kids, 7] & 0xFFFF)
                assert np.array_equal(a, ints[kids[order], 1]) and np.array_equal(cn, ints[kids[order], 6].astype(np.uint32))
                assert np.array_equal(cw.view(np.uint32), floats[kids[order], 0].view(np.uint32))
                assert np.array_equal(cp.view(np.uint32), floats[0, 1 + a].view(np.uint32))
        external = ply % 2 == 1
        alive = [g for g in range(games) if osp.game_alive(g)]
        if external:
            acts = np.full(games, -1, dtype=np.int32)
            for g in alive:
                acts[g] = int(rng.choice([c for c in range(hw) if c not in played[g]]))
            sp.set_actions(acts)
            osp.set_actions(acts)
            lit.set_actions(acts)
        else:
            acts = sp.sample_actions(1.0, 4)
            assert np.array_equal(acts, osp.sample(1.0, 4)) and np.array_equal(acts, lit.sample(1.0, 4))
        for g in alive:
            played[g].add(int(acts[g]))
        nm = sp.mirror_generate()
        om = osp.mirror_generate()
        lm, lmg = lit.mirror_generate()
        assert nm == len(om) and np.array_equal(sp.mirror_inputs(), om)
        pm = sp.mirror_eval()
        sp.mirror_apply()
        osp.advance(pm)
        lit.advance(pm[_perm(alive, lmg)], external=external)
        assert osp.error == 0 and lit.error == 0
        _dumps_equal(sp, osp, games, f"ply {ply}")
        for g in range(games):
            if lit.game_alive(g):
                for side in (0, 1):
                    li, lf = lit.tree_dump(g, side)
                    oi, of = osp.tree_dump(g, side)
                    assert np.array_equal(li[:, :7], oi[:, :7]) and np.array_equal(lf.view(np.uint32), of.view(np.uint32))
        if osp.alive_count == 0:
            break
    for g in range(games):  # transitions exist for the sampled plies only
        gb, gt, gp, gz = sp.replay(g)
        ob, ot, op, oz = osp.replay(g)
        lb, lt, lp, lz = lit.replay(g)
        assert len(gb) == len(ob) == len(lb)
        assert np.array_equal(gb, ob) and np.array_equal(gt, ot) and np.array_equal(gz, oz) and np.array_equal(gp.view(np.uint32), op.view(np.uint32))
        assert np.array_equal(gb, lb) and np.array_equal(gz, lz) and np.array_equal(gp.view(np.uint32), lp.view(np.uint32))
    eng.close()


def _perm(games_a, games_l):
    where, seen, out = {}, {}, []
    for r, g in enumerate(games_a):
        where.setdefault(int(g), []).append(r)
    for g in games_l:
        j = seen.get(int(g), 0)
        out.append(where[int(g)][j])
        seen[int(g)] = j + 1
    return np.array(out, dtype=np.int64)


@pytest.mark.parametrize("n,left", [(9, 1), (9, 3), (15, 1)])
def test_tree_level_draw(n, left):
    """A full board without an exact five, played in through omok_play_actions (one call per ply, the engine's own net for
    the mirror step) up to `left` empty cells; the search then expands Draw terminals inside the tree (ST_DRAW children,
    propagate(0.0), terminal-leaf revisits with z = 0; pme.rs:130-135,92-97) and the game ends as GameStatus::Draw."""
    hw = n * n
    seq = draw_sequence(n)
    eng, sp = _engine(n, 2, 8, seed=5)
    sp.reset()
    root_p = eng.evaluate_p(O.Environment(n).encode_nn_input(0)[None]).reshape(-1)
    osp = O.SelfPlay(n, 2, cap_nodes=1024, cap_tables=512, seed=5)
    osp.reset(root_p)
    for ply in range(hw - left):
        acts = np.array([seq[ply], seq[ply]], dtype=np.int32)
        osp.set_actions(acts)
        pm = eng.evaluate_p(osp.mirror_generate())  # the same rows omok_play_actions evaluates inside the engine
        sp.play_actions(acts)
        osp.advance(pm.reshape(len(pm), -1))
    _dumps_equal(sp, osp, 2, "after the external plies")
    saw_draw_node = False
    while osp.alive_count > 0:
        _search(sp, osp, 24, 8)
        _dumps_equal(sp, osp, 2, "after execute")
        for g in range(2):
            if osp.game_alive(g):
                ints, _ = osp.tree_dump(g, osp.ply & 1)
                saw_draw_node |= bool(np.any(ints[:, 2] == oa.api.DRAW))
        a = sp.sample_actions(1.0, 0)
        assert np.array_equal(a, osp.sample(1.0, 0))
        sp.mirror_generate()
        assert np.array_equal(sp.mirror_inputs(), osp.mirror_generate())
        pm = sp.mirror_eval()
        sp.mirror_apply()
        osp.advance(pm)
        _dumps_equal(sp, osp, 2, "after advance")
    alive, status, plies = sp.game_info()
    assert [int(s) for s in status] == [osp.game_status(g) for g in range(2)]
    if left == 1:
        assert saw_draw_node and np.all(status == oa.api.DRAW) and np.all(plies == hw)
        for g in range(2):
            b, t, pi, z = sp.replay(g)
            assert len(b) == 1 and z[0] == 0.0 and np.count_nonzero(pi[0]) == 1
    eng.close()


def test_play_actions_errors_leave_the_position_unchanged():
    eng, sp = _engine(9, 3, 8)
    sp.reset()
    sp.play_actions([40, 41, 42])
    before = [sp.tree_dump(g, s) for g in range(3) for s in (0, 1)]
    with pytest.raises(B.OmokError) as ei:  # game 1: occupied cell
        sp.play_actions([0, 41, 1])
    assert ei.value.code == -5
    with pytest.raises(B.OmokError) as ei:  # out of range
        sp.play_actions([0, 81, 1])
    assert ei.value.code == -5
    with pytest.raises(B.OmokError) as ei:  # a live game without a move
        sp.play_actions([0, -1, 1])
    assert ei.value.code == -1
    after = [sp.tree_dump(g, s) for g in range(3) for s in (0, 1)]
    for (ai, af), (bi, bf) in zip(before, after):
        assert np.array_equal(ai, bi) and np.array_equal(af, bf)
    assert sp.ply == 1
    sp.play_actions([0, 1, 2])
    assert sp.ply == 2
    eng.close()


@pytest.mark.parametrize("n", [9, 15])
def test_env_place_stone_on_caller_held_boards(n):
    hw = n * n
    eng = oa.Engine(board_size=n, games=1, max_nodes=16, max_tables=8, max_batch_k=1)
    rng = np.random.default_rng(1)
    b = 48
    boards = np.zeros((b, hw), dtype=np.uint8)
    turns = np.zeros(b, dtype=np.uint8)
    legal = np.full(b, hw, dtype=np.uint16)
    envs = [O.Environment(n) for _ in range(b)]
    seqs = [rng.permutation(hw) for _ in range(b)]
    seqs[0] = np.array(draw_sequence(n))
    for step in range(hw):
        acts = np.array([s[step] for s in seqs], dtype=np.int32)
        acts[1::7] = rng.integers(-2, hw + 3, size=len(acts[1::7]))  # out of range / already occupied cells too
        st = eng.env_place_stone(boards, turns, legal, acts)
        for i, e in enumerate(envs):
            a = int(acts[i])
            want = e.place_stone(a) if 0 <= a < hw else None
            assert st[i] == (-1 if want is None else want), (step, i)
            assert np.array_equal(boards[i], e.board) and turns[i] == e.turn and legal[i] == e.legal_move_count
    assert st[0] == oa.api.DRAW
    eng.close()


def test_every_reset_takes_a_new_rng_stream():
    """The reference draws fresh thread_rng values in every trainer iteration: reset number i runs on stream key
    seed + i * 0x9E3779B97F4A7C15 (same on the oracle), so consecutive episodes differ and a resumed run can be put on
    the stream of its iteration (omok_set_episode)."""
    n, games, count, k = 9, 3, 24, 8

    def play(sp, osp, plies=3):
        moves = []
        for _ in range(plies):
            _search(sp, osp, count, k)
            a = sp.sample_actions(1.0, 30)
            assert np.array_equal(a, osp.sample(1.0, 30))
            moves.append(a.copy())
            sp.mirror_generate()
            osp.mirror_generate()
            pm = sp.mirror_eval()
            sp.mirror_apply()
            osp.advance(pm)
        _dumps_equal(sp, osp, games, "episode")
        return np.stack(moves)

    eng, sp = _engine(n, games, k, seed=9)
    root_p = eng.evaluate_p(O.Environment(n).encode_nn_input(0)[None]).reshape(-1)
    osp = O.SelfPlay(n, games, cap_nodes=1024, cap_tables=512, seed=9)
    episodes = []
    for ep in range(3):
        sp.reset()
        osp.reset(root_p)
        episodes.append(play(sp, osp))
    assert not np.array_equal(episodes[0], episodes[1]) and not np.array_equal(episodes[1], episodes[2])
    sp.set_episode(1)
    osp.set_episode(1)
    sp.reset()
    osp.reset(root_p)
    assert np.array_equal(play(sp, osp), episodes[1])
    eng.close()


def test_replay_pack_is_deterministic_and_in_game_order():
    n, games, k = 9, 12, 8
    recs = []
    for _ in range(2):
        eng, sp = _engine(n, games, k, seed=2)
        sp.reset()
        sp.run(16, k)
        rec = sp.replay_record_bytes()
        _, _, plies = sp.game_info()
        total = int(plies.sum())
        buf = torch.full(((total + 3) * rec,), 0xAB, dtype=torch.uint8, device="cuda")  # dirty: pad bytes must be written
        assert sp.replay_pack_into(buf.data_ptr(), total + 3) == total
        host = buf.cpu().numpy()[: total * rec].reshape(total, rec)
        hw, brd = n * n, (n * n + 1 + 3) // 4 * 4
        i = 0
        for g in range(games):
            b, t, pi, z = sp.replay(g)
            for p in range(len(b)):
                r = host[i]
                assert np.array_equal(r[:hw], b[p]) and r[hw] == t[p] and np.all(r[hw + 1:brd] == 0)
                assert np.array_equal(r[brd:brd + 4 * hw].view(np.float32), pi[p]) and r[brd + 4 * hw:].view(np.float32)[0] == z[p]
                i += 1
        assert i == total
        recs.append(host.copy())
        eng.close()
    assert np.array_equal(recs[0], recs[1])


def test_state_and_argument_errors():
    with pytest.raises(B.OmokError) as ei:  # above OMOK_MAX_ARENA
        oa.Engine(board_size=9, games=1, max_nodes=20000, max_tables=64)
    assert ei.value.code == -1
    eng, sp = _engine(9, 2, 8)
    sp.reset()
    sp.round_generate(0, 8)
    with pytest.raises(B.OmokError) as ei:  # would overwrite the pending round's request counter / output rows
        eng.evaluate_pv(np.zeros((1, 3 * 81), dtype=np.float32))
    assert ei.value.code == -3
    sp.round_eval()
    sp.round_scatter()
    eng.evaluate_pv(np.zeros((1, 3 * 81), dtype=np.float32))
    big = oa.Engine(board_size=9, games=1, max_nodes=16384, max_tables=16384, max_batch_k=8)  # 82 KiB of re-rooting scratch: opt-in LDS
    big.load_random_weights(0)
    bsp = oa.SelfPlay(big)
    bsp.reset()
    bsp.execute(16, 8)
    bsp.sample_actions(1.0, 30)
    bsp.advance()
    big.close()
    eng.close()
