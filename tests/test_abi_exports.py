"""The C-ABI library loads here (no GPU) and exports every symbol include/mctq_hip.h declares."""
import ctypes
import os
import re

from conftest import REPO


def _declared_symbols():
    text = open(os.path.join(REPO, "include", "mctq_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mctq_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    from mct_quantizers_amd.hip import native
    assert _declared_symbols() == sorted(native.SIGNATURES)


def test_library_loads_and_exports_everything():
    from mct_quantizers_amd.hip import build, native
    path = build.build()                        # hipcc cross-compiles for gfx950 without a GPU
    lib = ctypes.CDLL(path)
    for name in _declared_symbols():
        assert hasattr(lib, name), name
    handle = native.load()
    assert handle.mctq_abi_version() == native.ABI_VERSION
    assert handle.mctq_last_error() == b""


def test_argument_validation_needs_no_gpu():
    from mct_quantizers_amd.hip import native
    lib = native.load()
    assert lib.mctq_fq_per_tensor_f32(None, None, -5, 1.0, 0, 0, 255, None) == native.MCTQ_E_ARG
    assert b"n < 0" in lib.mctq_last_error()
    assert lib.mctq_fq_per_tensor_f32(None, None, 0, 1.0, 0, 0, 255, None) == 0          # empty: nothing launched
    assert lib.mctq_fq_per_channel_f32(None, None, 0, 4, 8, None, None, 0, 255, None) == 0
    assert lib.mctq_lut_per_tensor_f32(None, None, 0, 1.0, 1.0, None, 4, 128.0, -128.0, 127.0, None) \
        == native.MCTQ_E_ARG                                                              # lut NULL
    assert lib.mctq_set_tuning(b"unroll", 3) == native.MCTQ_E_ARG
    assert lib.mctq_set_tuning(b"unroll", 4) == 0


def test_missing_library_is_loud(monkeypatch):
    import pytest
    from mct_quantizers_amd.hip import native
    monkeypatch.setattr(native, "_lib", None)
    monkeypatch.setenv("MCTQ_HIP_LIB", "/nonexistent/libmctq_hip.so")
    with pytest.raises(native.NativeLibraryError):
        native.load()
    assert not native.is_available()
