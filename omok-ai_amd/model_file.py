"""Host-side reader/writer of the reference's weights file (alpha-zero/src/model_io.rs:20-24,59-120: bincode 1.3.3 default
encoding of SavedData{variable_names: Vec<String>, parameters: Vec<Vec<f32>>}) for tools that hold tensors in host memory;
the engine itself reads and writes the same format through omok_net_load_file / omok_net_save_file."""
import struct

import numpy as np


def save(path, names, params):
    with open(path, "wb") as f:
        f.write(struct.pack("<Q", len(names)))
        for s in names:
            b = s.encode("utf-8")
            f.write(struct.pack("<Q", len(b)) + b)
        f.write(struct.pack("<Q", len(params)))
        for t in params:
            t = np.ascontiguousarray(t, dtype="<f4").ravel()
            f.write(struct.pack("<Q", t.size) + t.tobytes())


def load(path):
    with open(path, "rb") as f:
        data = f.read()
    pos = 0

    def take(n):
        nonlocal pos
        if pos + n > len(data):
            raise ValueError(f"{path}: unexpected end of file")
        pos += n
        return pos - n

    def u64():
        return struct.unpack_from("<Q", data, take(8))[0]

    names = []
    for _ in range(u64()):
        n = u64()
        o = take(n)
        names.append(data[o:o + n].decode("utf-8"))
    params = []
    for _ in range(u64()):
        n = u64()
        o = take(4 * n)
        params.append(np.frombuffer(data, dtype="<f4", count=n, offset=o).copy())
    return names, params
