"""C-ABI library: loads, exports every symbol the header declares, and its host-side
entry points (tile grid) agree with the reference fixtures.  CPU only: no kernel runs."""
import ctypes as C
import hashlib
import re
from pathlib import Path

import numpy as np
import pytest

REPO = Path(__file__).resolve().parents[1]


def declared_symbols():
    text = (REPO / "include" / "deephisto_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dh_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported_and_bound(built_lib):
    from deephisto_amd import _lib
    syms = declared_symbols()
    assert len(syms) >= 15
    for s in syms:
        assert hasattr(built_lib, s), f"{s} declared in include/deephisto_hip.h but not exported"
        assert s in _lib.SIGNATURES, f"{s} has no ctypes signature in deephisto_amd/_lib.py"
    assert sorted(_lib.SIGNATURES) == syms
    assert built_lib.dh_abi_version() == 1


def test_tile_grid_host_entry_matches_reference(built_lib, golden_meta, golden_grids):
    from deephisto_amd import tiles
    for name, g in golden_meta["grids"].items():
        o, n_unique = tiles.tile_grid(g["h"], g["w"], g["patch"], g["stride"], g["batch"])
        assert n_unique == g["n_unique"] and len(o) == g["n_padded"]
        assert hashlib.sha256(o.tobytes()).hexdigest() == g["sha256_int32_yx_padded"], name


def test_tile_grid_errors(built_lib):
    from deephisto_amd import _lib, tiles
    with pytest.raises(_lib.DeephistoHipError, match="smaller than patch"):
        tiles.tile_grid(100, 300, 256, 256, 4)
    with pytest.raises(_lib.DeephistoHipError):
        tiles.tile_grid(512, 512, 256, 0, 4)
    n = C.c_int64()
    out = np.zeros((2, 2), np.int32)
    rc = built_lib.dh_tile_grid(1024, 1024, 256, 256, 4, out.ctypes.data_as(C.c_void_p), 2)
    assert rc == -22 and b"capacity" in built_lib.dh_last_error()


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from deephisto_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", tmp_path / "nope.so")
    with pytest.raises(_lib.DeephistoHipError, match="no CPU fallback"):
        _lib.lib()
