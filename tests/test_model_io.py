"""Weights file of the reference (alpha-zero/src/model_io.rs:20-24,59-120; bincode 1.3.3 default encoding).

The reference has no saved model and no test for this path: parity is unpinned by the reference.  The oracle
(oracle/model_io.py) is pinned against bincode's published encoding by a hand-assembled byte string; the engine's C-ABI
entry points are then checked against the oracle in both directions."""
import os
import struct
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import model_io as M  # noqa: E402

# SavedData { variable_names: ["a", "bc"], parameters: [[1.0], [2.0, -3.5]] }, assembled by hand from the format:
# u64 LE lengths, UTF-8 bytes, f32 LE.
KAT = (
    struct.pack("<Q", 2) + struct.pack("<Q", 1) + b"a" + struct.pack("<Q", 2) + b"bc"
    + struct.pack("<Q", 2) + struct.pack("<Q", 1) + bytes.fromhex("0000803f")
    + struct.pack("<Q", 2) + bytes.fromhex("00000040") + bytes.fromhex("000060c0")
)


def test_oracle_writer_matches_hand_assembled_bytes(tmp_path):
    p = tmp_path / "kat.bin"
    M.model_save(p, ["a", "bc"], [np.array([1.0], np.float32), np.array([2.0, -3.5], np.float32)])
    assert p.read_bytes() == KAT
    assert len(KAT) == 8 + (8 + 1) + (8 + 2) + 8 + (8 + 4) + (8 + 8)


def test_oracle_reader_and_positional_assign(tmp_path):
    p = tmp_path / "kat.bin"
    p.write_bytes(KAT)
    names, params = M.model_load(p)
    assert names == ["a", "bc"]
    assert [t.tolist() for t in params] == [[1.0], [2.0, -3.5]]
    assert [t.tolist() for t in M.model_assign([1, 2], params)] == [[1.0], [2.0, -3.5]]
    assert len(M.model_assign([1], params)) == 1            # zip stops at the variables (model_io.rs:98)
    with pytest.raises(ValueError):
        M.model_assign([1, 2, 3], params)                    # a placeholder would stay unfed
    with pytest.raises(ValueError):
        M.model_assign([2, 2], params)                       # copy_from_slice length mismatch (model_io.rs:106)
    for cut in (3, 12, len(KAT) - 1):                        # truncated files
        p.write_bytes(KAT[:cut])
        with pytest.raises(ValueError):
            M.model_load(p)


def test_oracle_roundtrip_full_net(tmp_path):
    import omok_ai_amd as oa
    tensors = oa.weights.init_random(9, seed=4)
    p = tmp_path / "net9.bin"
    M.model_save(p, oa.weights.tensor_names(), tensors)
    names, params = M.model_load(p)
    assert names == oa.weights.tensor_names() and len(params) == 31
    for a, b in zip(tensors, params):
        assert np.array_equal(np.asarray(a, np.float32).ravel().view(np.uint32), b.view(np.uint32))
    assert os.path.getsize(p) == 8 + sum(8 + len(s.encode()) for s in names) + 8 + sum(8 + 4 * t.size for t in params)


@pytest.mark.gpu
def test_engine_save_and_load_match_the_oracle(tmp_path):
    import omok_ai_amd as oa
    from omok_ai_amd import binding as B
    n = 9
    tensors = oa.weights.init_random(n, seed=2)
    a = oa.Engine(board_size=n, games=4, max_nodes=16, max_tables=8, max_batch_k=8)
    a.load_weights(tensors)
    fa = tmp_path / "engine.bin"
    a.save(fa)                                             # ModelIO::save
    names, params = M.model_load(fa)
    assert names == oa.weights.tensor_names()
    for t, q in zip(tensors, params):
        assert np.array_equal(np.asarray(t, np.float32).ravel().view(np.uint32), q.view(np.uint32))
    fb = tmp_path / "oracle.bin"                            # ModelIO::load of a file written by the oracle, with TF-style
    M.model_save(fb, [f"whatever_{i}:0" for i in range(33)],  # names (ignored) and two surplus vectors (ignored)
                 list(tensors) + [np.zeros(3, np.float32), np.ones(1, np.float32)])
    b = oa.Engine(board_size=n, games=4, max_nodes=16, max_tables=8, max_batch_k=8)
    b.load(fb)
    x = (np.random.default_rng(0).random((40, 3 * n * n)) < 0.3).astype(np.float32)
    pa, va = a.evaluate_pv(x)
    pb, vb = b.evaluate_pv(x)
    assert np.array_equal(pa.view(np.uint32), pb.view(np.uint32)) and np.array_equal(va.view(np.uint32), vb.view(np.uint32))
    # error behaviour: missing file, truncated file, too few vectors, wrong length -- and the engine stays usable
    bad = tmp_path / "bad.bin"
    with pytest.raises(B.OmokError):
        b.load(tmp_path / "missing.bin")
    bad.write_bytes(fb.read_bytes()[:1000])
    with pytest.raises(B.OmokError):
        b.load(bad)
    M.model_save(bad, ["x"] * 30, list(tensors)[:30])
    with pytest.raises(B.OmokError):
        b.load(bad)
    wrong = list(tensors)
    wrong[5] = np.zeros(7, np.float32)
    M.model_save(bad, ["x"] * 31, wrong)
    with pytest.raises(B.OmokError):
        b.load(bad)
    pb2, vb2 = b.evaluate_pv(x)                             # a rejected file leaves the loaded net untouched
    assert np.array_equal(pb2.view(np.uint32), pb.view(np.uint32)) and np.array_equal(vb2.view(np.uint32), vb.view(np.uint32))
    c = oa.Engine(board_size=15, games=4, max_nodes=16, max_tables=8, max_batch_k=8)
    with pytest.raises(B.OmokError):                        # a 9x9 file into a 15x15 net: fc0_w / p_fc0 lengths differ
        c.load(fb)
    with pytest.raises(B.OmokError):                        # nothing loaded yet: nothing to save
        c.save(tmp_path / "none.bin")
    c.close()
    a.close()
    b.close()
