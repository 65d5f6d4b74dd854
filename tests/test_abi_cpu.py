"""The C-ABI library loads without a GPU and exports every symbol include/isegmi.h declares; host-only
entry points (weight packing, descriptor validation, error reporting) behave; nothing computes on a GPU here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "isegmi.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(isegmi_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from isegmi import _ffi
    lib = _ffi.lib()
    syms = declared_symbols()
    assert len(syms) >= 40
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, missing
    assert lib.isegmi_version() == 1


def test_missing_library_fails_loudly(monkeypatch):
    from isegmi import _ffi
    monkeypatch.setattr(_ffi, "_lib", None)
    monkeypatch.setattr(_ffi, "LIB_PATH", "/nonexistent/libisegmi.so")
    with pytest.raises(_ffi.IsegmiError, match="no CPU fallback"):
        _ffi.lib()


def test_pack_weights_layout_and_validation():
    from isegmi import _ffi
    d = _ffi.make_conv_desc(1, 8, 8, 32, 5, 1, 1)
    w = np.arange(5 * 32, dtype=np.float32).reshape(5, 1, 1, 32)
    p = _ffi.pack_conv_weights(d, w).reshape(128, 32)   # Cout padded to 128 rows
    assert not p[5:].any()
    # inside each group of 8 consecutive k: [k0 k2 k4 k6 | k1 k3 k5 k7]
    assert list(p[1, :8]) == [32, 34, 36, 38, 33, 35, 37, 39]
    assert sorted(p[3]) == list(range(96, 128))
    stem = _ffi.make_conv_desc(1, 32, 32, 4, 64, 7, 7, 2, 3)
    ps = _ffi.pack_conv_weights(stem, np.ones((64, 7, 7, 4), np.float32)).reshape(128, 7, 32)
    assert ps[0].sum() == 7 * 28 and ps[0, 0].sum() == 28  # 28 real taps + 4 zero pads per kernel row
    assert _ffi.conv_out_hw(stem) == (16, 16)
    bad = _ffi.make_conv_desc(1, 8, 8, 24, 5, 1, 1)  # Cin not a multiple of 32
    with pytest.raises(_ffi.IsegmiError, match="Cin"):
        _ffi.conv_out_hw(bad)
    assert b"Cin" in _ffi.lib().isegmi_last_error()


def test_no_device_is_an_error_not_a_fallback():
    from isegmi import _ffi
    if _ffi.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(_ffi.IsegmiError):
        _ffi.DeviceBuffer((16,))
