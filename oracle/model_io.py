"""CPU restatement of the reference's weights file (TEST INFRASTRUCTURE ONLY; see oracle/oracle.py for who may import).

alpha-zero/src/model_io.rs:
  :20-24   struct SavedData { variable_names: Vec<String>, parameters: Vec<Vec<f32>> }
  :59-90   ModelIO::save  -> bincode::serialize_into(file, &saved_data)
  :92-120  ModelIO::load  -> bincode::deserialize_from(file); zip(variables, parameters): POSITIONAL, names unused (:98);
                             Tensor::copy_from_slice (:106) panics on a length mismatch.
The encoding lives in a third-party dependency that is not in the reference tree: bincode 1.3.3 (Cargo.lock), default
options of the free functions serialize_into / deserialize_from = little-endian, FIXED-width integers:
  Vec<T>  = u64 length, then the elements;   String = u64 byte length, then the UTF-8 bytes;   f32 = 4 bytes LE.
PARITY UNPINNED BY THE REFERENCE: it ships no saved model and no test of this path.  Pinned here against bincode's
published encoding by a hand-assembled byte string (tests/test_model_io.py).
"""
import struct

import numpy as np


def model_save(path, names, params):
    """ModelIO::save (model_io.rs:59-90)."""
    with open(path, "wb") as f:
        f.write(struct.pack("<Q", len(names)))
        for s in names:
            b = s.encode("utf-8")
            f.write(struct.pack("<Q", len(b)) + b)
        f.write(struct.pack("<Q", len(params)))
        for t in params:
            t = np.ascontiguousarray(t, dtype="<f4").ravel()
            f.write(struct.pack("<Q", t.size) + t.tobytes())


def model_load(path):
    """deserialize_from of model_io.rs:94: returns (variable_names, parameters) exactly as stored."""
    with open(path, "rb") as f:
        data = f.read()
    pos = 0

    def u64():
        nonlocal pos
        if pos + 8 > len(data):
            raise ValueError("unexpected end of file")
        (v,) = struct.unpack_from("<Q", data, pos)
        pos += 8
        return v

    names = []
    for _ in range(u64()):
        n = u64()
        if pos + n > len(data):
            raise ValueError("unexpected end of file")
        names.append(data[pos:pos + n].decode("utf-8"))
        pos += n
    params = []
    for _ in range(u64()):
        n = u64()
        if pos + 4 * n > len(data):
            raise ValueError("unexpected end of file")
        params.append(np.frombuffer(data, dtype="<f4", count=n, offset=pos).copy())
        pos += 4 * n
    return names, params


def model_assign(sizes, params):
    """The zip of ModelIO::load (model_io.rs:98-108): positional; extra parameters ignored; a missing one leaves a
    placeholder unfed (session.run fails) and a wrong length panics: both are errors here."""
    if len(params) < len(sizes):
        raise ValueError("fewer parameter vectors than variables")
    out = []
    for size, t in zip(sizes, params):
        if t.size != size:
            raise ValueError("parameter length differs from the variable's element count")
        out.append(t)
    return out
