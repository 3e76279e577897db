"""CPU-side checks of the drop-in boundary: the shared library builds for
gfx950, loads, and exports every symbol include/pisa_hip.h declares.  No
compute call is made (no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g

    g.build()
    from pisa_amd import _lib

    assert os.path.exists(_lib.LIB_PATH)
    return _lib


def test_header_symbols_all_exported(built):
    header = open(os.path.join(ROOT, "include", "pisa_hip.h")).read()
    declared = set(re.findall(r"\b(pisa_hip_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 25
    handle = ctypes.CDLL(built.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(handle, name), "symbol %s missing from libpisa_hip.so" % name
    # the binding covers the whole header
    assert declared == set(built.EXPORTED_SYMBOLS)


def test_struct_layout_matches_header(built):
    # sizes implied by include/pisa_hip.h (LP64)
    assert ctypes.sizeof(built.Prob3Params) == (9 + 18 + 18 + 18 + 9) * 8 + 8
    assert ctypes.sizeof(built.Earth) == 8 + 8 + 3 * 64 * 8
    assert ctypes.sizeof(built.Binning) == 8 + 3 * 8 * 3
    assert ctypes.sizeof(built.Container) == 8 + 5 * 8 + 3 * 8 + 5 * 8 + 4 + 4 + 8 + 8 + 2 * 8


def test_status_strings_and_no_gpu_behaviour(built):
    lib = built.lib()
    assert lib.pisa_hip_strerror(0) == b"ok"
    assert b"120 layers" in lib.pisa_hip_strerror(-2)
    assert lib.pisa_hip_version() >= 100
    # argument validation happens before any device access
    p = built.Prob3Params()
    assert lib.pisa_hip_propagate_array(p, 1, None, None, None, 10, 200, 1, None, None) == -2
    assert lib.pisa_hip_propagate_array(p, 0, None, None, None, 10, 4, 1, None, None) == -1
    assert lib.pisa_hip_grid_plan_destroy(None) == 0


def test_missing_library_fails_loudly(monkeypatch, built):
    monkeypatch.setattr(built, "_lib", None)
    monkeypatch.setattr(built, "LIB_PATH", "/nonexistent/libpisa_hip.so")
    with pytest.raises(ImportError):
        built.lib()
