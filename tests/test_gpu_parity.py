"""GPU parity tests (run with -m gpu on an MI355X).  Everything goes through the C ABI.

 - rules: the reference's 8 environment tests (environment/src/lib.rs:201-426) on the device kernel
 - net: evaluate_pv vs the oracle's fp32 restatement, tolerance 1e-3 (BASELINE.json north_star)
 - tree search: bit-exact canonical tree dumps, request boards, sampled moves and replay tuples vs
   the oracle, with the GPU net's (p, v) fed to both sides so the comparison isolates the tree
   arithmetic (select / expand / backup / noise / sampling / re-rooting)
"""
import os

import numpy as np
import pytest

import omok_ai_amd as oa
from omok_ai_amd import binding as B
from oracle import oracle as O
from helpers import draw_sequence, random_positions, tree_shape

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
TOL = 1e-3  # north_star: policy/value within 1e-3 of the reference CPU path


@pytest.fixture(scope="module", params=[9, 15])
def eng_env(request):
    e = oa.Engine(board_size=request.param, games=1, max_nodes=64, max_tables=32, max_batch_k=1)
    yield e
    e.close()


# ---- environment crate known answers ----------------------------------------------------------
def test_place_stone(eng_env):  # lib.rs:201-252
    env = oa.Environment(eng_env)
    assert env.turn == oa.api.TURN_BLACK
    for i in range(12):
        assert env.place_stone(i) == oa.api.IN_PROGRESS
        assert env.board[i] == (oa.api.BLACK if i % 2 == 0 else oa.api.WHITE)
        assert env.turn == (oa.api.TURN_WHITE if i % 2 == 0 else oa.api.TURN_BLACK)
    assert env.place_stone(3) is None
    assert env.legal_move_count == eng_env.hw - 12


def test_game_endings(eng_env):  # lib.rs:255-372
    n = eng_env.n
    horiz = [x + r * n for x in range(4) for r in (0, 1)] + [4]
    vert = [c + y * n for y in range(4) for c in (0, 2)] + [4 * n]
    fill = list(range(n * 4))
    moves = np.full((4, n * 4 + 1), -1, dtype=np.int32)
    moves[0, :9], moves[1, :9] = horiz, vert
    moves[2, :] = fill + [n * 4 + 4]
    moves[3, :] = fill + [n * 4]
    st, boards, turns, legal = eng_env.env_play(moves)
    assert list(st[0, :9]) == [0] * 8 + [oa.api.BLACK_WIN]
    assert list(st[1, :9]) == [0] * 8 + [oa.api.BLACK_WIN]
    assert st[2, -1] == oa.api.BLACK_WIN and st[3, -1] == oa.api.BLACK_WIN
    assert np.all(st[0, 9:] == -1)  # index -1 is rejected like Option::None


def test_rules_match_oracle_on_random_games(eng_env):
    n, hw = eng_env.n, eng_env.hw
    rng = np.random.default_rng(5)
    moves = np.stack([rng.permutation(hw) for _ in range(64)]).astype(np.int32)
    moves[::2, 7] = moves[::2, 3]  # an occupied cell -> None (these rows can never fill the board)
    moves[1] = draw_sequence(n)    # a full board without an exact five: the last move is GameStatus::Draw
    st, boards, turns, legal = eng_env.env_play(moves)
    assert st[1, -1] == oa.api.DRAW and np.all(st[1, :-1] == oa.api.IN_PROGRESS) and legal[1] == 0
    for b in range(moves.shape[0]):
        env = O.Environment(n)
        for i, m in enumerate(moves[b]):
            s = env.place_stone(int(m))
            assert (s if s is not None else -1) == st[b, i], (b, i)
        assert np.array_equal(env.board, boards[b]) and env.turn == turns[b] and env.legal_move_count == legal[b]


def test_encodings(eng_env):  # lib.rs:375-426 + encoder.rs:10-46
    n, hw = eng_env.n, eng_env.hw
    env = oa.Environment(eng_env)
    for i in (0, 10, 2, 30):
        env.place_stone(i)
    exp = np.zeros(2 * hw, dtype=np.float32)
    exp[[0, 10 * 2 + 1, 2 * 2, 30 * 2 + 1]] = 1.0
    assert np.array_equal(env.encode_board(oa.api.TURN_BLACK), exp)
    exp = np.zeros(2 * hw, dtype=np.float32)
    exp[[1, 10 * 2, 2 * 2 + 1, 30 * 2]] = 1.0
    assert np.array_equal(env.encode_board(oa.api.TURN_WHITE), exp)
    oenv = O.Environment(n)
    for i in (0, 10, 2):
        oenv.place_stone(i)
    for mode in (0, 1):
        got = eng_env.encode_nn_input(oenv.board[None], np.array([oenv.turn], dtype=np.uint8), mode).reshape(-1)
        assert np.array_equal(got, oenv.encode_nn_input(mode))


# ---- net ----------------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [9, 15])
@pytest.mark.parametrize("mode", [B.NET_F16X3, B.NET_F32])
def test_net_parity(n, mode):
    g = np.load(os.path.join(GOLD, f"net_n{n}.npz"))
    tensors = oa.weights.init_random(n, seed=int(g["seed"]))
    eng = oa.Engine(board_size=n, games=8, max_nodes=16, max_tables=8, max_batch_k=16, net_mode=mode)
    eng.load_weights(tensors)
    rng = np.random.default_rng(3)
    extra = []
    for _ in range(150):  # more positions than the fixture, incl. a batch that is not a multiple of the tile
        env = O.Environment(n)
        for c in rng.permutation(n * n)[: int(rng.integers(0, n * n - 1))]:
            env.place_stone(int(c))
        extra.append(env.encode_nn_input(int(rng.integers(0, 2))))
    inputs = np.concatenate([g["inputs"], np.stack(extra)])
    p, v = eng.evaluate_pv(inputs)
    p, v = p.reshape(len(inputs), -1), v.reshape(-1)
    k = len(g["inputs"])
    assert np.abs(p[:k] - g["p"]).max() < TOL and np.abs(v[:k] - g["v"]).max() < TOL  # vs torch float64 fixture
    pc, vc = O.Net(n, tensors).forward(inputs, threads=8)
    dp, dv = np.abs(p - pc).max(), np.abs(v - vc).max()
    print(f"n={n} mode={mode}: max|dp|={dp:.3e} max|dv|={dv:.3e}")
    assert dp < TOL and dv < TOL
    if mode == B.NET_F16X3:
        assert dp < 8e-4 and dv < 8e-4, "split-precision net should be ~1e-4 (worst case seen 5e-4 at N=9); larger means lost correction terms"
    assert np.allclose(p.sum(axis=1), 1.0, atol=1e-4)
    eng.close()


_random_positions = random_positions  # (tests/helpers.py)


def test_net_ragged_batches_are_row_independent():
    """A row's result must not depend on what else is in the batch (empty, 1, tile-1, tile+1 rows): bit-identical."""
    n = 15
    eng = oa.Engine(board_size=n, games=32, max_nodes=16, max_tables=8, max_batch_k=16)
    eng.load_random_weights(0)
    x = _random_positions(n, 300, 11)
    p, v = eng.evaluate_pv(x)
    for b in (1, 31, 127, 128, 129, 257):
        pb, vb = eng.evaluate_pv(x[:b])
        assert np.array_equal(pb.view(np.uint32), p[:b].view(np.uint32)) and np.array_equal(vb.view(np.uint32), v[:b].view(np.uint32)), b
    eng.close()


@pytest.mark.parametrize("games,rows", [(2048, 32768), (1100, 17600)])
def test_net_large_batch_kernel_paths(games, rows):
    """fc0 picks its split-K factor from the launch size: 32768 rows = 256 tiles run unsplit (k_fc0_mx<EPI_SPLIT>, the path
    of full self-play rounds), 17600 rows = 137.5 tiles run 3-way split with a half-full last tile.  Same results as the
    15-way split of a small engine (different fp32 summation order in front of saturated softmaxes: 2e-4 allowed, 4e-5
    seen), and within TOL of the oracle."""
    n = 15
    small = oa.Engine(board_size=n, games=32, max_nodes=16, max_tables=8, max_batch_k=16)
    big = oa.Engine(board_size=n, games=games, max_nodes=8, max_tables=4, max_batch_k=16)
    tensors = oa.weights.init_random(n, seed=0)
    small.load_weights(tensors)
    big.load_weights(tensors)
    base = _random_positions(n, 400, 5)
    x = np.tile(base, (rows // 400 + 1, 1))[:rows]  # full capacity in one launch
    pb, vb = big.evaluate_pv(x)
    ps, vs = small.evaluate_pv(base)
    pb, ps = pb.reshape(len(x), -1), ps.reshape(len(base), -1)
    vb, vs = vb.reshape(-1), vs.reshape(-1)
    for r in (1, rows // 400 - 1):  # every copy of the base positions gives the same rows
        assert np.array_equal(pb[r * 400:(r + 1) * 400].view(np.uint32), pb[:400].view(np.uint32))
    assert np.abs(pb[:400] - ps).max() < 2e-4 and np.abs(vb[:400] - vs).max() < 2e-4
    pc, vc = O.Net(n, tensors).forward(base[:96], threads=8)
    assert np.abs(pb[:96] - pc).max() < TOL and np.abs(vb[:96] - vc).max() < TOL
    small.close()
    big.close()


def test_evaluate_logits_is_the_same_forward():
    """omok_evaluate_logits returns what sits in front of the Softmax / Tanh ops of the same forward."""
    n = 15
    x = _random_positions(n, 200, 4)
    for mode in (B.NET_F16X3, B.NET_F32):
        eng = oa.Engine(board_size=n, games=16, max_nodes=16, max_tables=8, max_batch_k=16, net_mode=mode)
        eng.load_random_weights(0)
        p, v = eng.evaluate_pv(x)
        lg, vp = eng.evaluate_logits(x)
        e = np.exp((lg - lg.max(axis=1, keepdims=True)).astype(np.float64))
        sm = e / e.sum(axis=1, keepdims=True)
        assert np.abs(sm - p.reshape(len(x), -1)).max() < 2e-6
        assert np.abs(np.tanh(vp.astype(np.float64)) - v.reshape(-1)).max() < 2e-6
        eng.close()


# ---- self-play: tree arithmetic bit-exact ---------------------------------------------------------
@pytest.mark.parametrize("n,k,games,path", [(15, 16, 40, "copy"), (15, 16, 448, "difference"), (9, 8, 40, "copy"), (9, 8, 320, "difference")])
def test_search_round_outputs_on_the_sibling_path(n, k, games, path):
    """The p / v a SEARCH ROUND produces against the fp32 kernels and against the row-by-row path (evaluate_pv) of the same
    engine, on the very request rows of the rounds.  Rounds of >= 3072 rows (N = 15; 1024 at N = 9) take the difference path (base row + window difference
    rows, DESIGN 3.3: different rounding, inside the contract), smaller rounds the copy path (bit-identical to row-by-row)."""
    count = 6 * k
    tensors = oa.weights.init_random(n, seed=3)
    eng = oa.Engine(board_size=n, games=games, max_nodes=512, max_tables=128, max_batch_k=k, seed=11, net_mode=B.NET_F16X3)
    eng.load_weights(tensors)
    ref = oa.Engine(board_size=n, games=games, max_nodes=8, max_tables=4, max_batch_k=k, net_mode=B.NET_F32)
    ref.load_weights(tensors)
    sp = oa.SelfPlay(eng)
    sp.reset()
    rows, dp, dv, dpp, dvp, differing = 0, 0.0, 0.0, 0.0, 0.0, 0
    for ply in range(3 if path == "difference" else 4):
        for rnd in range(count // k):
            nreq = sp.round_generate(rnd, k, 0.25, 0.03)
            x = sp.round_inputs().copy()
            p, v = sp.round_eval()
            assert len(x) == nreq and len(p) == nreq
            p, v = np.array(p).reshape(nreq, -1).copy(), np.array(v).reshape(-1).copy()
            sp.round_scatter()
            if rnd == 0:
                continue  # (the first round of a ply has one request per tree: no siblings)
            assert nreq == games * k
            p32, v32 = ref.evaluate_pv(x)
            pp, vp = eng.evaluate_pv(x)
            p32, pp = p32.reshape(nreq, -1), pp.reshape(nreq, -1)
            dp, dv = max(dp, np.abs(p - p32).max()), max(dv, np.abs(v - v32.reshape(-1)).max())
            dpp, dvp = max(dpp, np.abs(p - pp).max()), max(dvp, np.abs(v - vp.reshape(-1)).max())
            differing += int((p.view(np.uint32) != pp.view(np.uint32)).any(axis=1).sum())
            rows += nreq
        sp.sample_actions(1.0, 30)
        sp.mirror_generate()
        sp.mirror_eval()
        sp.mirror_apply()
    print(f"{path} path, {rows} rows: vs fp32 max|dp| {dp:.2e} max|dv| {dv:.2e}; vs row-by-row max|dp| {dpp:.2e} max|dv| {dvp:.2e}, {differing} rows differ")
    assert dp < 1e-3 and dv < 1e-3
    if path == "copy":
        assert differing == 0 and dvp == 0.0
    else:
        assert differing > rows // 2     # (the path under test really ran)
        assert dpp < 5e-4 and dvp < 5e-4
    eng.close()
    ref.close()


def test_rows_mode_is_row_independent_in_search_rounds():
    """OMOK_NET_F16X3_ROWS: a search round's p / v are bit-identical to evaluate_pv of the same rows (no sibling differences)."""
    n, games, k = 15, 12, 16
    eng = oa.Engine(board_size=n, games=games, max_nodes=512, max_tables=128, max_batch_k=k, seed=2, net_mode=B.NET_F16X3_ROWS)
    eng.load_random_weights(1)
    sp = oa.SelfPlay(eng)
    sp.reset()
    for rnd in range(4):
        nreq = sp.round_generate(rnd, k, 0.25, 0.03)
        x = sp.round_inputs().copy()
        p, v = sp.round_eval()
        p, v = np.array(p).reshape(nreq, -1).copy(), np.array(v).reshape(-1).copy()
        sp.round_scatter()
        pp, vp = eng.evaluate_pv(x)
        assert np.array_equal(p.view(np.uint32), pp.reshape(nreq, -1).view(np.uint32))
        assert np.array_equal(v.view(np.uint32), vp.reshape(-1).view(np.uint32))
    eng.close()


def _compare_trees(sp, osp, games, tag):
    for g in range(games):
        for side in (0, 1):
            gi, gf = sp.tree_dump(g, side)
            oi, of = osp.tree_dump(g, side)
            assert gi.shape == oi.shape, f"{tag}: game {g} side {side}: {gi.shape[0]} vs {oi.shape[0]} nodes"
            assert np.array_equal(gi, oi), f"{tag}: node records differ (game {g} side {side})"
            assert np.array_equal(gf.view(np.uint32), of.view(np.uint32)), f"{tag}: w/policy bits differ (game {g} side {side})"
            assert sp.tree_root(g, side)[0] == osp.tree_root(g, side)[0]
            assert np.float32(sp.tree_root(g, side)[1]).tobytes() == np.float32(osp.tree_root(g, side)[1]).tobytes()


def _drive_selfplay_vs_oracle(n, games, count, k, max_plies, mode, threshold=6, max_nodes=2048, max_tables=1024, seed=7,
                              game_offset=5, weight_seed=0):
    """Plays `games` games step by step on the engine and on the oracle (the oracle consumes the GPU net's p / v, so the
    comparison isolates the tree arithmetic) and compares canonical tree dumps, request boards, moves and replay tuples bit
    for bit.  Returns the shape of the deepest search seen by the ORACLE: (fully expanded nodes, fully expanded non-root
    nodes, max depth, max tables) so that a test can prove it left the shallow root-plus-one-layer regime."""
    tensors = oa.weights.init_random(n, seed=weight_seed)
    eng = oa.Engine(board_size=n, games=games, max_nodes=max_nodes, max_tables=max_tables, max_batch_k=k, seed=seed,
                    game_offset=game_offset, net_mode=mode)
    eng.load_weights(tensors)
    sp = oa.SelfPlay(eng)
    sp.reset()
    root_p = eng.evaluate_p(O.Environment(n).encode_nn_input(0)[None]).reshape(-1)
    osp = O.SelfPlay(n, games, cap_nodes=max_nodes, cap_tables=max_tables, seed=seed, game_offset=game_offset)
    osp.reset(root_p)
    _compare_trees(sp, osp, games, "reset")
    shape = [0, 0, 0, 0]
    ply = 0
    rounds = (count + k - 1) // k
    while osp.alive_count > 0 and (max_plies == 0 or ply < max_plies):
        for rnd in range(rounds):
            nreq = sp.round_generate(rnd, k, 0.25, 0.03)
            oin = osp.round_generate(rnd, k, 0.25, 0.03)
            assert nreq == len(oin), f"ply {ply} round {rnd}: request count"
            assert np.array_equal(sp.round_inputs(), oin), f"ply {ply} round {rnd}: request boards"
            if rnd == 0 or (rounds > 20 and rnd % 16 == 15):  # pending children of the round are in the dump as well
                _compare_trees(sp, osp, games, f"ply {ply} after generate of round {rnd}")
            p, v = sp.round_eval()
            sp.round_scatter()
            osp.round_scatter(p, v)
        _compare_trees(sp, osp, games, f"ply {ply} after execute")
        assert osp.error == 0
        for g in range(games):
            if osp.game_alive(g):
                full, full_nr, depth = tree_shape(osp.tree_dump(g, ply & 1)[0])
                shape = [max(shape[0], full), max(shape[1], full_nr), max(shape[2], depth), max(shape[3], osp.tree_root(g, ply & 1)[3])]
        a = sp.sample_actions(1.0, threshold)
        assert np.array_equal(a, osp.sample(1.0, threshold)), f"ply {ply}: actions"
        nm = sp.mirror_generate()
        om = osp.mirror_generate()
        assert nm == len(om) and np.array_equal(sp.mirror_inputs(), om)
        pm = sp.mirror_eval()
        sp.mirror_apply()
        osp.advance(pm)
        _compare_trees(sp, osp, games, f"ply {ply} after advance")
        alive, status, plies = sp.game_info()
        assert [int(x) for x in alive] == [osp.game_alive(g) for g in range(games)]
        assert [int(x) for x in status] == [osp.game_status(g) for g in range(games)]
        ply += 1
    assert sp.alive_count == osp.alive_count
    for g in range(games):
        gb, gt, gp, gz = sp.replay(g)
        ob, ot, op, oz = osp.replay(g)
        assert np.array_equal(gb, ob) and np.array_equal(gt, ot) and np.array_equal(gz, oz)
        assert np.array_equal(gp.view(np.uint32), op.view(np.uint32))
    eng.close()
    return tuple(shape)


@pytest.mark.parametrize("n,games,count,k,max_plies,mode", [
    (9, 6, 48, 8, 0, B.NET_F16X3),     # whole games to the end on the reference's board size
    (9, 3, 40, 16, 12, B.NET_F32),     # count not a multiple of K (rounds up), fp32 net path
    (15, 3, 64, 16, 6, B.NET_F16X3),   # the benchmark board, shallow
    (9, 3, 64, 32, 0, B.NET_F16X3),    # K = 32: batches of more than 16 children (two passes of the 16-at-a-time win checks), whole games
    (9, 2, 128, 64, 0, B.NET_F16X3),   # K = KMAX = 64: a batch as wide as the wave (the sorted ranks of the picks fill every lane)
])
def test_selfplay_bit_exact_vs_oracle(n, games, count, k, max_plies, mode):
    _drive_selfplay_vs_oracle(n, games, count, k, max_plies, mode)


# The regime the headline number is measured in (BASELINE.json configs): 800 / 1600 simulations per move at N = 15 reach
# fully expanded nodes (the 225-wide PUCT scan of k_round<15>, IT = 4), depth >= 3 (deep backup), several child tables
# (multi-table transition<15>) and the select-leaf memo.  The asserts on the oracle's tree shape make sure these tests
# cannot silently fall back to the root-plus-one-layer regime of the small configs above.
@pytest.mark.parametrize("n,games,count,k,max_plies,min_full,min_full_nonroot,min_depth,min_tables", [
    (15, 2, 800, 16, 4, 1, 1, 3, 3),    # configs[1]/[3] per-tree workload (C2/C4): first 4 plies
    (15, 1, 1600, 16, 2, 1, 1, 3, 3),   # configs[4] (C5): 1600 simulations per move
    (9, 4, 200, 8, 0, 1, 1, 3, 3),      # configs[2] (C3): 9x9, 200 simulations, K = 8, whole games
])
def test_selfplay_bit_exact_benchmark_regime(n, games, count, k, max_plies, min_full, min_full_nonroot, min_depth, min_tables):
    cap = min(16384, 4 * count + 1024)
    full, full_nr, depth, tables = _drive_selfplay_vs_oracle(n, games, count, k, max_plies, B.NET_F16X3, threshold=30,
                                                              max_nodes=cap, max_tables=max(256, cap // 4))
    print(f"oracle tree shape: {full} fully expanded nodes ({full_nr} non-root), depth {depth}, {tables} tables")
    assert full >= min_full and full_nr >= min_full_nonroot and depth >= min_depth and tables >= min_tables


def test_selfplay_single_game_c1_whole_game():
    """BASELINE.json configs[0] on the engine: ONE 15x15 game, 100 simulations per move (-> 112 with K = 16), the
    reference's mode rule (Boltzmann for 30 plies, then Best), played to the end and compared with the oracle at every
    step (games = 1: grid of one wave, every scan / compaction kernel at its smallest size)."""
    _drive_selfplay_vs_oracle(15, 1, 100, 16, 0, B.NET_F16X3, threshold=30, max_nodes=4096, max_tables=1024, seed=0, game_offset=0)


def test_execute_matches_stepwise_and_is_deterministic():
    """omok_execute (all rounds enqueued without host round trips) == the step-wise path."""
    n, games, count, k = 9, 5, 32, 8
    tensors = oa.weights.init_random(n, seed=0)
    dumps = []
    for variant in ("execute", "stepwise", "execute"):
        eng = oa.Engine(board_size=n, games=games, max_nodes=1024, max_tables=512, max_batch_k=k, seed=3)
        eng.load_weights(tensors)
        sp = oa.SelfPlay(eng)
        sp.reset()
        for _ in range(3):
            if variant == "execute":
                sp.execute(count, k)
                sp.sample_actions(1.0, 2)
                sp.advance()
            else:
                for rnd in range(count // k):
                    sp.round_generate(rnd, k)
                    sp.round_eval()
                    sp.round_scatter()
                sp.sample_actions(1.0, 2)
                sp.mirror_generate()
                sp.mirror_eval()
                sp.mirror_apply()
        dumps.append([sp.tree_dump(g, s) for g in range(games) for s in (0, 1)])
        eng.close()
    for a, b in ((0, 1), (0, 2)):
        for (ai, af), (bi, bf) in zip(dumps[a], dumps[b]):
            assert np.array_equal(ai, bi) and np.array_equal(af.view(np.uint32), bf.view(np.uint32))


def test_selfplay_run_whole_episode_properties():
    """Size-independent properties of a full episode through omok_selfplay_run."""
    n, games, count, k = 9, 64, 32, 16
    eng = oa.Engine(board_size=n, games=games, max_nodes=1024, max_tables=512, max_batch_k=k, seed=1)
    eng.load_random_weights(0)
    sp = oa.SelfPlay(eng)
    sp.reset()
    st = sp.run(count, k)
    assert sp.alive_count == 0 and st["finished"] == games
    alive, status, plies = sp.game_info()
    assert np.all(alive == 0) and np.all(status != 0)
    assert st["ply_games"] == plies.sum()
    assert st["sims"] == plies.sum() * count
    for g in range(0, games, 7):
        boards, turns, pi, z = sp.replay(g)
        assert len(boards) == plies[g]
        assert np.all((boards != 0).sum(axis=1) == np.arange(len(boards)))
        assert np.allclose(pi.sum(axis=1), 1.0, atol=1e-5)
        assert np.all(z[:-1] == 0) and z[-1] == (1.0 if status[g] >= 2 else 0.0)
        # the recorded game replays to the recorded status under the rules kernel
        moves = []
        for i in range(len(boards) - 1):
            moves.append(int(np.flatnonzero(boards[i + 1] != boards[i])[0]))
        s, _, _, _ = eng.env_play(np.array([moves], dtype=np.int32))
        assert np.all(s == 0)
    eng.close()


def test_error_paths():
    eng = oa.Engine(board_size=9, games=2, max_nodes=8, max_tables=4, max_batch_k=8)
    sp = oa.SelfPlay(eng)
    with pytest.raises(B.OmokError):  # net not loaded
        sp.reset()
    eng.load_random_weights(0)
    with pytest.raises(B.OmokError):  # reset not called
        sp.execute(8, 8)
    sp.reset()
    with pytest.raises(B.OmokError):  # batch_size above max_batch_k
        sp.execute(8, 16)
    with pytest.raises(B.OmokError) as ei:  # arena of 8 nodes overflows
        sp.execute(64, 8)
    assert ei.value.code == -4
    eng.close()
    with pytest.raises(B.OmokError):
        oa.Engine(board_size=10, games=1)
