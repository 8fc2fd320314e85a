"""The reference's own 8 `environment` tests (environment/src/lib.rs:201-426), transliterated as
known-answer tests for the oracle, at the reference's N=9 and at N=15, plus derived rule cases
 and the 5 symmetry tests of src/utils.rs:70-108."""
import numpy as np
import pytest

from oracle import oracle as O


@pytest.mark.parametrize("n", [9, 15])
def test_place_stone(n):  # lib.rs:201-252
    env = O.Environment(n)
    assert env.turn == O.TURN_BLACK
    for i in range(12):
        assert env.place_stone(i) == O.IN_PROGRESS
        assert env.board[i] == (O.BLACK if i % 2 == 0 else O.WHITE)
        assert env.turn == (O.TURN_WHITE if i % 2 == 0 else O.TURN_BLACK)
    assert env.place_stone(3) is None  # occupied -> None (lib.rs:105-107)
    assert env.legal_move_count == n * n - 12


@pytest.mark.parametrize("n", [9, 15])
def test_game_ending_horizontal(n):  # lib.rs:255-298
    env = O.Environment(n)
    for x in range(4):
        assert env.place_stone(x + 0 * n) == O.IN_PROGRESS
        assert env.place_stone(x + 1 * n) == O.IN_PROGRESS
    assert env.place_stone(4 + 0 * n) == O.BLACK_WIN


@pytest.mark.parametrize("n", [9, 15])
def test_game_ending_vertical(n):  # lib.rs:301-344
    env = O.Environment(n)
    for y in range(4):
        assert env.place_stone(0 + y * n) == O.IN_PROGRESS
        assert env.place_stone(2 + y * n) == O.IN_PROGRESS
    assert env.place_stone(0 + 4 * n) == O.BLACK_WIN


@pytest.mark.parametrize("n", [9, 15])
def test_game_ending_lt_rb(n):  # lib.rs:347-358
    env = O.Environment(n)
    for index in range(n * 4):
        env.place_stone(index)
    assert env.place_stone(n * 4 + 4) == O.BLACK_WIN


@pytest.mark.parametrize("n", [9, 15])
def test_game_ending_lb_rt(n):  # lib.rs:361-372
    env = O.Environment(n)
    for index in range(n * 4):
        env.place_stone(index)
    assert env.place_stone(n * 4) == O.BLACK_WIN


@pytest.mark.parametrize("n", [9, 15])
def test_encoding_0(n):  # lib.rs:375-387
    env = O.Environment(n)
    env.place_stone(0)
    expected = np.zeros(2 * n * n, dtype=np.float32)
    expected[0] = 1.0
    assert np.array_equal(env.encode_board(O.TURN_BLACK), expected)


@pytest.mark.parametrize("n", [9, 15])
def test_encoding_1_and_2(n):  # lib.rs:389-426
    env = O.Environment(n)
    for i in (0, 10, 2, 30):
        env.place_stone(i)
    expected = np.zeros(2 * n * n, dtype=np.float32)
    expected[[0 * 2 + 0, 10 * 2 + 1, 2 * 2 + 0, 30 * 2 + 1]] = 1.0
    assert np.array_equal(env.encode_board(O.TURN_BLACK), expected)
    expected = np.zeros(2 * n * n, dtype=np.float32)
    expected[[0 * 2 + 1, 10 * 2 + 0, 2 * 2 + 1, 30 * 2 + 0]] = 1.0
    assert np.array_equal(env.encode_board(O.TURN_WHITE), expected)


@pytest.mark.parametrize("n", [9, 15])
def test_exactly_five_rule(n):
    """Derived from lib.rs:151-154 (== 5): six in a row is not a win; white wins too."""
    env = O.Environment(n)
    # black: cells 0,1,2, 4,5 on row 0; white on row 2; then black fills 3 -> six in a row
    blacks, whites = [0, 1, 2, 4, 5], [2 * n + i for i in range(5)]
    for b, w in zip(blacks, whites[:4]):
        assert env.place_stone(b) == O.IN_PROGRESS
        assert env.place_stone(w) == O.IN_PROGRESS
    assert env.place_stone(blacks[4]) == O.IN_PROGRESS
    assert env.place_stone(whites[4]) == O.WHITE_WIN  # white five
    env2 = O.Environment(n)
    for b, w in zip(blacks, [3 * n + 2 * i for i in range(5)]):
        env2.place_stone(b)
        env2.place_stone(w)
    assert env2.place_stone(3) == O.IN_PROGRESS  # XXX_XX + X = 6 -> not a win


def test_draw_when_board_full():
    n = 5
    env = O.Environment(n)
    # a 5x5 fill order without any five: pattern rows XXOXX / OOXOO ... play cells so nobody gets 5
    pattern = ["XXOXX", "OOXOO", "XXOXX", "OOXOO", "XOXOX"]
    xs = [r * n + c for r in range(n) for c in range(n) if pattern[r][c] == "X"]
    os_ = [r * n + c for r in range(n) for c in range(n) if pattern[r][c] == "O"]
    assert len(xs) == 13 and len(os_) == 12
    status = None
    for i in range(25):
        status = env.place_stone(xs[i // 2] if i % 2 == 0 else os_[i // 2])
        if i < 24:
            assert status == O.IN_PROGRESS, i
    assert status == O.DRAW and env.legal_move_count == 0


@pytest.mark.parametrize("n", [9, 15])
def test_encode_nn_input_layout(n):
    """encoder.rs:22-43: interleaved (mine, theirs) pairs then the turn plane."""
    env = O.Environment(n)
    for i in (0, 10, 2):
        env.place_stone(i)  # black 0,2 white 10; white to move
    hw = n * n
    f = env.encode_nn_input(O.MODE_PLAYER)
    exp = np.zeros(3 * hw, dtype=np.float32)
    exp[[0 * 2 + 1, 2 * 2 + 1, 10 * 2 + 0]] = 1.0  # perspective = white
    assert np.array_equal(f, exp)  # turn plane 0: white to move
    f = env.encode_nn_input(O.MODE_OPPONENT)
    exp = np.zeros(3 * hw, dtype=np.float32)
    exp[[0 * 2 + 0, 2 * 2 + 0, 10 * 2 + 1]] = 1.0
    assert np.array_equal(f, exp)
    env.place_stone(5)
    assert np.all(env.encode_nn_input(O.MODE_PLAYER)[2 * hw:] == 1.0)  # black to move


def _sym(fn, src, size=2):
    import ctypes as C
    src = np.asarray(src, dtype=np.float32)
    dst = np.zeros_like(src)
    getattr(O.lib(), fn)(src.ctypes.data_as(C.POINTER(C.c_float)), dst.ctypes.data_as(C.POINTER(C.c_float)), size)
    return dst.tolist()


def test_symmetry_helpers():  # src/utils.rs:70-108
    assert _sym("orc_rotate_90", [1, 2, 3, 4]) == [3, 1, 4, 2]
    assert _sym("orc_rotate_180", [1, 2, 3, 4]) == [4, 3, 2, 1]
    assert _sym("orc_rotate_270", [1, 2, 3, 4]) == [2, 4, 1, 3]
    assert _sym("orc_flip_horizontal", [1, 2, 3, 4]) == [2, 1, 4, 3]
    assert _sym("orc_flip_vertical", [1, 2, 3, 4]) == [3, 4, 1, 2]
