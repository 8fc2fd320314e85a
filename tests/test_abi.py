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
