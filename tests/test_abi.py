"""C-ABI library: loads, exports every symbol the header declares, and its host-side
entry points (tile grid) agree with the reference fixtures.  CPU only: no kernel runs."""
import ctypes as C
import hashlib
import re
from pathlib import Path

import numpy as np
import pytest

REPO = Path(__file__).resolve().parents[1]


def declared_symbols(header="deephisto_hip.h"):
    text = (REPO / "include" / header).read_text()
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
    assert not [s for s in syms if "debug" in s], "test hooks belong in include/deephisto_hip_debug.h, not in the boundary header"


def test_debug_header_symbols_exported_and_bound(built_lib):
    """The per-kernel test hooks live in their own, unversioned header: still exported, still bound."""
    from deephisto_amd import _lib
    syms = declared_symbols("deephisto_hip_debug.h")
    assert len(syms) >= 20 and all("debug" in s for s in syms)
    for s in syms:
        assert hasattr(built_lib, s), f"{s} declared in include/deephisto_hip_debug.h but not exported"
    assert sorted(_lib.DEBUG_SIGNATURES) == syms
    assert not set(_lib.DEBUG_SIGNATURES) & set(_lib.SIGNATURES)


def test_library_exports_exactly_the_two_headers(built_lib):
    """`nm -D` of the shared object: every exported dh_* symbol is declared in one of the two headers, and vice versa."""
    import shutil
    import subprocess
    from deephisto_amd import _lib
    nm = shutil.which("nm") or "/opt/rocm/lib/llvm/bin/llvm-nm"
    out = subprocess.run([nm, "-D", "--defined-only", str(_lib.LIB_PATH)], capture_output=True, text=True, check=True).stdout
    exported = sorted({ln.split()[-1] for ln in out.splitlines() if ln.split() and ln.split()[-1].startswith("dh_")})
    assert exported == sorted(declared_symbols() + declared_symbols("deephisto_hip_debug.h"))


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


def test_tile_grid_property_against_the_pinned_oracle(built_lib):
    """Random ragged geometries (non-divisible sizes, stride != patch, every padding length incl. none): the host entry of the C ABI ==
    oracle/tiling.py, which is pinned to the reference's own `_create_batched_coords` output by the fixtures above (full_samplers.py:331-352)."""
    from hypothesis import given, settings, strategies as st
    from deephisto_amd import tiles
    from oracle import tiling

    @settings(max_examples=200, deadline=None)
    @given(st.integers(1, 64), st.integers(1, 64), st.integers(0, 700), st.integers(0, 700), st.integers(1, 9))
    def check(patch8, stride8, eh, ew, batch):
        patch, stride = 8 * patch8, 8 * stride8
        h, w = patch + eh, patch + ew
        want = tiling.batched_origins(h, w, patch, stride, batch).reshape(-1, 2)   # the C entry returns the padded list, un-batched
        got, n_unique = tiles.tile_grid(h, w, patch, stride, batch)
        assert n_unique == len(tiling.tile_origins(h, w, patch, stride))
        assert got.dtype == np.int32 and got.shape == want.shape and np.array_equal(got, want)

    check()


def test_no_ablation_switch_in_shipped_library(built_lib):
    """the timing-only ablations (DH_T2_ABL: whole kernel groups skipped, results garbage) and the gradient dump
    (DH_TRAIN_DUMP) are compile-time macros of diagnostic builds; the shipped shared object does not know the names."""
    from deephisto_amd import _lib
    blob = _lib.LIB_PATH.read_bytes()
    for name in (b"DH_T2_ABL", b"DH_TRAIN_DUMP", b"DH_ABL", b"SP_ABL"):
        assert name not in blob, f"{name.decode()} is compiled into the shipped library"


def _env_table(built_lib):
    buf = C.create_string_buffer(1 << 16)
    assert built_lib.dh_debug_env_knobs(buf, len(buf)) == 0
    rows = [ln.split() for ln in buf.value.decode().splitlines()]
    return {r[0]: (int(r[1]), int(r[2]), int(r[3]), r[4]) for r in rows}


def test_env_knobs_are_one_table_and_documented(built_lib):
    """Every DH_* name the shared object contains is a row of csrc/env_knobs.h (or DH_RCCL_LIB, a path), and INTEGRATION.md lists
    every row with the same default and range."""
    from deephisto_amd import _lib
    table = _env_table(built_lib)
    assert len(table) >= 10
    in_binary = set(re.findall(rb"\x00(DH_[A-Z0-9_]{3,})(?=\x00)", _lib.LIB_PATH.read_bytes()))   # whole C strings (what a getenv would take)
    names = {n.decode() for n in in_binary} - {"DH_RCCL_LIB", "DH_OK", "DH_EINVAL", "DH_ENOMEM", "DH_EHIP"}
    assert names <= set(table), f"environment names outside env_knobs.h: {sorted(names - set(table))}"
    doc = (REPO / "INTEGRATION.md").read_text()
    for name, (default, lo, hi, read) in table.items():
        m = re.search(rf"\| `{name}` \| (-?\d+) \| (-?\d+) \.\. (-?\d+) \| (\w+) \|", doc)
        assert m, f"{name} is not documented in INTEGRATION.md"
        assert (int(m.group(1)), int(m.group(2)), int(m.group(3)), m.group(4)) == (default, lo, hi, read), name
    assert "`DH_RCCL_LIB`" in doc


def test_unrecognised_env_value_is_refused_by_name(built_lib, monkeypatch):
    """A mistyped switch is not atoi'd: the create entry points fail before any GPU call and dh_last_error() names the variable."""
    h = C.c_void_p()
    for name, bad in (("DH_T2_FOLD", "yes"), ("DH_CONV_S2_WIDE", "2"), ("DH_G2_NS3_K", "12x"), ("DH_T2_SIDE", "")):
        monkeypatch.setenv(name, bad)
        assert built_lib.dh_train2_create(C.byref(h), b"resnet50", 5) == -22
        assert name.encode() in built_lib.dh_last_error()
        assert built_lib.dh_resnet18_create(C.byref(h), 5, 1) == -22
        assert name.encode() in built_lib.dh_last_error()
        monkeypatch.delenv(name)


def test_rccl_entry_never_loads_a_library_of_its_own(built_lib, monkeypatch):
    """dh_allgather_logits only ever calls into an RCCL that is already in the process: a handle without ncclAllGather is
    refused, and so is an exchange in a process that has loaded no RCCL at all."""
    libc = C.CDLL("libc.so.6")
    assert built_lib.dh_set_rccl(C.c_void_p(libc._handle)) == -22 and b"ncclAllGather" in built_lib.dh_last_error()
    assert built_lib.dh_set_rccl(None) == 0
    monkeypatch.setenv("DH_RCCL_LIB", "/nonexistent/librccl.so.1")
    dummy = (C.c_float * 8)()
    comm = C.c_void_p(1)
    assert built_lib.dh_allgather_logits(comm, dummy, dummy, 1, 5, None) == -22
    assert b"not loaded in this process" in built_lib.dh_last_error()
    assert built_lib.dh_set_rccl(None) == 0
