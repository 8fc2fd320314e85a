"""No-GPU checks of the boundary: the C-ABI library loads and exports every symbol that
include/omok_mi355x.h declares; creating an engine without a GPU fails loudly (no CPU path)."""
import ctypes as C
import os
import re

import pytest

import omok_ai_amd  # noqa: F401
from omok_ai_amd import binding

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "omok_mi355x.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(omok_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert _declared_symbols() == sorted(binding.SYMBOLS)


def test_library_exports_every_declared_symbol():
    assert os.path.exists(binding.LIB_PATH), "run __graft_entry__.build() first"
    lib = C.CDLL(binding.LIB_PATH)
    for sym in _declared_symbols():
        assert hasattr(lib, sym), sym


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(binding.OmokError) as ei:
        omok_ai_amd.Engine(board_size=9, games=1)
    assert "no CPU path" in str(ei.value) or "HIP" in str(ei.value)


def test_product_never_touches_the_oracle():
    pkg = os.path.join(ROOT, "omok-ai_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.lower().replace("not the oracle", ""), f


# ---- the Rust side of the boundary (bindings/omok_mi355x.rs): rustc is absent, so the text is compared with the header ----
def _abi_text():
    import importlib.util
    spec = importlib.util.spec_from_file_location("abi_text", os.path.join(ROOT, "tools", "abi_text.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_rust_binding_declares_the_whole_header():
    A = _abi_text()
    c, rs = A.parse_header(), A.parse_rust()
    assert sorted(c) == _declared_symbols(), "the prototype parser must see every symbol of the header"
    assert sorted(rs) == sorted(c), (sorted(set(c) - set(rs)), sorted(set(rs) - set(c)))
    for name, (ret, args) in c.items():
        rret, rargs = rs[name]
        assert rret == ret, (name, "return", ret, rret)
        assert len(rargs) == len(args), (name, "arity")
        for (an, at), (rn, rt) in zip(args, rargs):
            assert rt == at, (name, an, at, rt)  # kind, width, pointer depth and constness


def test_rust_config_struct_and_constants_match_the_header():
    A = _abi_text()
    assert [(n, t) for n, t in A.parse_rust_config()] == [(n, t) for n, t in A.parse_header_config()]
    # ... and the ctypes mirror the tests call through has the same layout
    kinds = {C.c_int32: ("int", 32), C.c_uint64: ("uint", 64), C.c_int64: ("int", 64)}
    assert [(n, kinds[t] + ((),)) for n, t in binding.Config._fields_] == A.parse_header_config()
    assert A.parse_rust_consts() == A.parse_defines()
    assert A.parse_defines()["OMOK_STAT_COUNT"] == len(binding.STAT_NAMES)
