"""ctypes binding of libmctq_hip.so (C ABI declared in include/mctq_hip.h).

The handle lives in a module global, never on a quantizer object, so quantizers and the
modules that hold them stay picklable (torch.save(model) is how MCT ships models,
reference pytorch/load_model.py:23-34).

There is no fallback: if the shared library is missing or its ABI version does not match,
every call raises.  Build it with ``python -m mct_quantizers_amd.hip.build`` (or
``__graft_entry__.build()``).
"""
from __future__ import annotations

import ctypes
import os
import threading

ABI_VERSION = 9
DT_F32, DT_F16, DT_BF16, DT_F64 = 0, 1, 2, 3
CODE_I8, CODE_U8, CODE_I4, CODE_U4 = 0, 1, 2, 3
FQ_ITEM_PER_TENSOR = 1
LIB_NAME = "libmctq_hip.so"
LIB_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lib")
MCTQ_E_ARG = -10001

_lock = threading.RLock()
_lib = None
TRACE = os.environ.get("MCTQ_ROCTX", "0") not in ("", "0")     # one roctx range per launch (see the end of the file)

_c_f32p = ctypes.c_void_p     # device pointers travel as integers
_c_i32p = ctypes.c_void_p


class FqItem(ctypes.Structure):
    """mctq_fq_item of include/mctq_hip.h (one tensor of a batched launch)."""
    _fields_ = [("x", ctypes.c_void_p), ("y", ctypes.c_void_p),
                ("outer", ctypes.c_int64), ("channels", ctypes.c_int64), ("inner", ctypes.c_int64),
                ("scales", ctypes.c_void_p), ("zero_points", ctypes.c_void_p),
                ("quant_min", ctypes.c_int32), ("quant_max", ctypes.c_int32),
                ("dtype", ctypes.c_int32), ("flags", ctypes.c_int32)]


class LutItem(ctypes.Structure):
    """mctq_lut_item of include/mctq_hip.h (one tensor of a batched LUT launch)."""
    _fields_ = [("x", ctypes.c_void_p), ("y", ctypes.c_void_p),
                ("outer", ctypes.c_int64), ("channels", ctypes.c_int64), ("inner", ctypes.c_int64),
                ("thresholds", ctypes.c_void_p), ("table", ctypes.c_void_p),
                ("entries", ctypes.c_int32), ("eps", ctypes.c_float),
                ("thr_div", ctypes.c_float), ("thr_mul", ctypes.c_float),
                ("mult", ctypes.c_float), ("clip_min", ctypes.c_float), ("clip_max", ctypes.c_float),
                ("dtype", ctypes.c_int32), ("step_round", ctypes.c_int32)]


# name -> (restype, argtypes); must list every symbol of include/mctq_hip.h
SIGNATURES = {
    "mctq_abi_version": (ctypes.c_int, []),
    "mctq_build_id": (ctypes.c_char_p, []),
    "mctq_last_error": (ctypes.c_char_p, []),
    "mctq_last_launch": (ctypes.c_char_p, []),
    "mctq_launch_count": (ctypes.c_int64, []),
    "mctq_set_tuning": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_int32]),
    "mctq_selftest_division": (ctypes.c_int, [_c_f32p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p]),
    "mctq_selftest_reciprocal": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    "mctq_fq_per_tensor_f32": (ctypes.c_int, [_c_f32p, _c_f32p, ctypes.c_int64, ctypes.c_float, ctypes.c_int32,
                                              ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p]),
    "mctq_fq_per_channel_f32": (ctypes.c_int, [_c_f32p, _c_f32p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                               _c_f32p, _c_i32p, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p]),
    "mctq_lut_per_tensor_f32": (ctypes.c_int, [_c_f32p, _c_f32p, ctypes.c_int64, ctypes.c_float, ctypes.c_float,
                                               _c_f32p, ctypes.c_int32, ctypes.c_float, ctypes.c_float,
                                               ctypes.c_float, ctypes.c_void_p]),
    "mctq_lut_per_channel_f32": (ctypes.c_int, [_c_f32p, _c_f32p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                                _c_f32p, ctypes.c_float, _c_f32p, ctypes.c_int32, ctypes.c_float,
                                                ctypes.c_float, ctypes.c_float, ctypes.c_void_p]),
    "mctq_fq_per_tensor": (ctypes.c_int, [_c_f32p, _c_f32p, ctypes.c_int64, ctypes.c_int32, ctypes.c_float, ctypes.c_int32,
                                          ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p]),
    "mctq_fq_per_channel": (ctypes.c_int, [_c_f32p, _c_f32p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                           ctypes.c_int32, _c_f32p, _c_i32p, ctypes.c_int32, ctypes.c_int32,
                                           ctypes.c_void_p]),
    "mctq_fq_per_tensor_tqp": (ctypes.c_int, [_c_f32p, _c_f32p, ctypes.c_int64, ctypes.c_int32, _c_f32p, _c_i32p,
                                              ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p]),
    "mctq_fq_batched": (ctypes.c_int, [ctypes.POINTER(FqItem), ctypes.c_int32, ctypes.c_void_p]),
    "mctq_fq_batch_pack": (ctypes.c_int64, [ctypes.POINTER(FqItem), ctypes.c_int32, ctypes.c_void_p, ctypes.c_int64]),
    "mctq_fq_batch_run": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "mctq_lutt_batch_pack": (ctypes.c_int64, [ctypes.POINTER(LutItem), ctypes.c_int32, ctypes.c_void_p, ctypes.c_int64]),
    "mctq_lutt_batch_run": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "mctq_lut_per_tensor_f64": (ctypes.c_int, [_c_f32p, _c_f32p, ctypes.c_int64, ctypes.c_double, ctypes.c_float,
                                               _c_f32p, ctypes.c_int32, ctypes.c_float, ctypes.c_float,
                                               ctypes.c_float, ctypes.c_void_p]),
    "mctq_fq_codes_per_tensor": (ctypes.c_int, [_c_f32p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32,
                                                ctypes.c_float, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                                ctypes.c_void_p]),
    "mctq_fq_codes_per_channel": (ctypes.c_int, [_c_f32p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64,
                                                 ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, _c_f32p, _c_i32p,
                                                 ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p]),
    "mctq_lut_per_tensor": (ctypes.c_int, [_c_f32p, _c_f32p, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32,
                                           ctypes.c_float, ctypes.c_float, _c_f32p, ctypes.c_int32, ctypes.c_float,
                                           ctypes.c_float, ctypes.c_float, ctypes.c_void_p]),
    "mctq_lut_per_channel": (ctypes.c_int, [_c_f32p, _c_f32p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                            ctypes.c_int32, _c_f32p, ctypes.c_float, _c_f32p, ctypes.c_int32,
                                            ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_void_p]),
    "mctq_lutt_per_tensor": (ctypes.c_int, [_c_f32p, _c_f32p, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32,
                                            ctypes.c_float, ctypes.c_float, _c_f32p, ctypes.c_int32, ctypes.c_float,
                                            ctypes.c_float, ctypes.c_float, ctypes.c_void_p]),
    "mctq_lutt_per_channel": (ctypes.c_int, [_c_f32p, _c_f32p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                             ctypes.c_int32, _c_f32p, ctypes.c_float, _c_f32p, ctypes.c_int32,
                                             ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_void_p]),
    "mctq_grid_per_tensor_f32": (ctypes.c_int, [_c_f32p, _c_f32p, ctypes.c_int64, ctypes.c_float, ctypes.c_float,
                                                ctypes.c_float, ctypes.c_int32, ctypes.c_void_p]),
    "mctq_grid_per_channel_f32": (ctypes.c_int, [_c_f32p, _c_f32p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                                 _c_f32p, _c_f32p, _c_f32p, ctypes.c_int32, ctypes.c_void_p]),
    "mctq_qlinear_i8": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_float, ctypes.c_void_p,
                                       _c_f32p, _c_i32p, _c_f32p, _c_f32p, ctypes.c_int64, ctypes.c_int64,
                                       ctypes.c_int64, ctypes.c_void_p]),
    "mctq_qlinear_i8_codes": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_float,
                                             ctypes.c_void_p, _c_f32p, _c_i32p, _c_f32p, ctypes.c_void_p, ctypes.c_int32,
                                             ctypes.c_float, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                             ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p]),
    "mctq_qlinear_w4a8": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_float,
                                         ctypes.c_void_p, _c_f32p, _c_i32p, _c_f32p, ctypes.c_void_p, ctypes.c_int32,
                                         ctypes.c_float, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                         ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p]),
    "mctq_fq_codes_nchw_to_nhwc": (ctypes.c_int, [_c_f32p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                                  ctypes.c_int32, ctypes.c_int32, ctypes.c_float, ctypes.c_int32,
                                                  ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p]),
    "mctq_lut_table_entries": (ctypes.c_int32, [ctypes.c_float, ctypes.c_float]),
    "mctq_lut_build_table": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_float, ctypes.c_float,
                                            ctypes.c_float, ctypes.c_void_p]),
    "mctq_lut_steps_f64_bytes": (ctypes.c_int32, [ctypes.c_int32]),
    "mctq_lut_build_steps_f64": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_float, ctypes.c_float,
                                                ctypes.c_float, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int32)]),
    "mctq_luts_per_tensor_f64": (ctypes.c_int, [_c_f32p, _c_f32p, ctypes.c_int64, ctypes.c_double, ctypes.c_float,
                                                ctypes.c_void_p, ctypes.c_int32, ctypes.c_float, ctypes.c_float,
                                                ctypes.c_float, ctypes.c_void_p]),
    "mctq_luts_per_channel_f64": (ctypes.c_int, [_c_f32p, _c_f32p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                                 _c_f32p, ctypes.c_float, ctypes.c_void_p, ctypes.c_int32,
                                                 ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_void_p]),
    "mctq_lut_steps_words": (ctypes.c_int32, [ctypes.c_int32]),
    "mctq_lut_build_steps": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_float, ctypes.c_float,
                                            ctypes.c_float, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int32)]),
    "mctq_luts_per_tensor": (ctypes.c_int, [_c_f32p, _c_f32p, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32,
                                            ctypes.c_float, ctypes.c_float, _c_f32p, ctypes.c_int32, ctypes.c_float,
                                            ctypes.c_float, ctypes.c_float, ctypes.c_void_p]),
    "mctq_luts_per_channel": (ctypes.c_int, [_c_f32p, _c_f32p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                             ctypes.c_int32, _c_f32p, ctypes.c_float, _c_f32p, ctypes.c_int32,
                                             ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_void_p]),
    "mctq_lutt_per_tensor_f32": (ctypes.c_int, [_c_f32p, _c_f32p, ctypes.c_int64, ctypes.c_float, ctypes.c_float,
                                                _c_f32p, ctypes.c_int32, ctypes.c_float, ctypes.c_float,
                                                ctypes.c_float, ctypes.c_void_p]),
    "mctq_lutt_per_channel_f32": (ctypes.c_int, [_c_f32p, _c_f32p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                                 _c_f32p, ctypes.c_float, _c_f32p, ctypes.c_int32, ctypes.c_float,
                                                 ctypes.c_float, ctypes.c_float, ctypes.c_void_p]),
}


class NativeLibraryError(RuntimeError):
    pass


def lib_path() -> str:
    return os.environ.get("MCTQ_HIP_LIB", os.path.join(LIB_DIR, LIB_NAME))


def _check_build_id(path: str, got: str, which: str):
    """A binary in lib/ must have been built from the sources that are in the tree now (content hash, hip/build.py);
    a stale one is refused, never used silently.  Libraries given through MCTQ_HIP_LIB are the caller's business;
    MCTQ_SKIP_BUILD_ID_CHECK=1 turns the check off (bisecting with hand-built binaries)."""
    if "MCTQ_HIP_LIB" in os.environ or os.environ.get("MCTQ_SKIP_BUILD_ID_CHECK", "0") not in ("", "0"):
        return
    from mct_quantizers_amd.hip import build as _build
    try:
        want = getattr(_build, which)()
    except OSError:                     # a deployment without the kernel sources next to the binary: nothing to compare with
        return
    if got != want:
        raise NativeLibraryError(f"{path} was built from other sources (build id {got}, the tree's is {want}); "
                                 f"rebuild it: python -m mct_quantizers_amd.hip.build")


def load():
    """Load (once) and return the ctypes handle; raises NativeLibraryError if unavailable."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        path = lib_path()
        if not os.path.exists(path):
            raise NativeLibraryError(
                f"{path} not found: the HIP kernels are not built. Run `python -m mct_quantizers_amd.hip.build` "
                f"(needs hipcc, --offload-arch=gfx950). mct_quantizers_amd has no fallback for GPU tensors.")
        try:
            handle = ctypes.CDLL(path)
        except OSError as e:  # pragma: no cover
            raise NativeLibraryError(f"cannot load {path}: {e}") from e
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(handle, name)
            except AttributeError as e:
                raise NativeLibraryError(f"{path} does not export {name}") from e
            fn.restype = res
            fn.argtypes = args
        got = handle.mctq_abi_version()
        if got != ABI_VERSION:
            raise NativeLibraryError(f"{path} has ABI version {got}, expected {ABI_VERSION}; rebuild it")
        _check_build_id(path, handle.mctq_build_id().decode(), "tree_build_id")
        # deployment knob: outputs up to this many MiB are stored through the caches (see include/mctq_hip.h)
        mb = os.environ.get("MCTQ_CACHED_STORE_MAX_MB")
        if mb:
            handle.mctq_set_tuning(b"cached_store_max_mb", int(mb))
        _lib = handle
    return _lib


def build_lut_steps_f64(lut_values, mult: float, clip_min: float, clip_max: float):
    """Host-side DOUBLE threshold list for float64 tensors (numpy uint8 blob + P; include/mctq_hip.h:
    mctq_lut_build_steps_f64), or None when the codebook does not qualify (the literal double scan is used then)."""
    import numpy as np
    lib = load()
    lut = np.ascontiguousarray(np.asarray(lut_values, dtype=np.float32).reshape(-1))
    if lut.size < 1 or lut.size > 4096:
        return None
    cap = lib.mctq_lut_steps_f64_bytes(lut.size)
    if cap <= 0:
        return None
    blob = np.zeros(cap, dtype=np.uint8)
    p = ctypes.c_int32(0)
    rc = lib.mctq_lut_build_steps_f64(lut.ctypes.data, lut.size, mult, clip_min, clip_max, blob.ctypes.data, ctypes.byref(p))
    if rc != 0:
        return None
    return blob[: p.value * 12 + 8].copy(), int(p.value)


def build_lut_steps(lut_values, mult: float, clip_min: float, clip_max: float):
    """Host-side threshold list for an integer codebook of any clip range (numpy float32 [2 P + 2], see
    include/mctq_hip.h: mctq_lut_build_steps), or None when the codebook does not qualify."""
    import numpy as np
    lib = load()
    lut = np.ascontiguousarray(np.asarray(lut_values, dtype=np.float32).reshape(-1))
    cap = lib.mctq_lut_steps_words(lut.size)
    if cap < 0:
        return None
    steps = np.zeros(cap, dtype=np.float32)
    n_words = ctypes.c_int32(0)
    rc = lib.mctq_lut_build_steps(lut.ctypes.data, lut.size, mult, clip_min, clip_max, steps.ctypes.data, ctypes.byref(n_words))
    if rc != 0:
        return None
    return steps[: n_words.value].copy()


# ------------------------------------------------------------------------------------------
# compiled binding of the hot entry points (csrc/binding/mctq_torch.cpp -> lib/_mctq_torch.so)
# ------------------------------------------------------------------------------------------
FAST_NAME = "_mctq_torch"
_fast = None
_fast_tried = False


def fast_path() -> str:
    return os.path.join(LIB_DIR, FAST_NAME + ".so")


def fast():
    """The compiled CPython binding over the same C ABI, or None.

    MCTQ_BINDING = "auto" (default): use it when it loads, otherwise the ctypes binding (both end in the same
    extern "C" entry points of libmctq_hip.so -- neither is a fallback away from the HIP kernels);
    "compiled": raise if it cannot be loaded; "ctypes": never use it.  With MCTQ_ROCTX=1 the ctypes binding is
    used so that every launch gets its roctx range."""
    global _fast, _fast_tried
    if _fast_tried:
        return _fast
    with _lock:
        if _fast_tried:
            return _fast
        mode = os.environ.get("MCTQ_BINDING", "auto")
        if mode == "ctypes" or TRACE:
            _fast_tried = True
            return None
        try:
            load()                                   # libmctq_hip.so first: the binding links against it
            import importlib.util
            import torch  # noqa: F401  (the module links libtorch_python; torch must be imported first)
            path = fast_path()
            spec = importlib.util.spec_from_file_location(FAST_NAME, path)
            if spec is None or not os.path.exists(path):
                raise ImportError(f"{path} not found")
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
            if mod.abi_version() != ABI_VERSION:
                raise ImportError(f"{path} was built against ABI {mod.abi_version()}, expected {ABI_VERSION}")
            _check_build_id(path, mod.build_id(), "binding_build_id")
            _fast = mod
        except Exception as e:  # noqa: BLE001
            if mode == "compiled":
                raise NativeLibraryError(f"compiled binding unavailable: {e}") from e
            import warnings
            warnings.warn(f"mct_quantizers_amd: compiled binding not loaded ({e}); using the ctypes binding of "
                          f"libmctq_hip.so (same kernels, ~2 us more host time per call). "
                          f"Build it with `python -m mct_quantizers_amd.hip.build`.")
            _fast = None
        _fast_tried = True
    return _fast


def is_available() -> bool:
    try:
        load()
        return True
    except NativeLibraryError:
        return False


def check(rc: int, what: str):
    if rc != 0:
        msg = load().mctq_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{what} failed (rc={rc}): {msg}")


def last_launch() -> str:
    """Kernel variant of this thread's last elementwise launch (see include/mctq_hip.h: mctq_last_launch)."""
    return load().mctq_last_launch().decode("utf-8", "replace")


def launch_count() -> int:
    """Kernel launches this thread has enqueued through the library so far (include/mctq_hip.h: mctq_launch_count)."""
    return int(load().mctq_launch_count())


def set_tuning(key: str, value: int):
    check(load().mctq_set_tuning(key.encode(), int(value)), f"mctq_set_tuning({key}={value})")


def build_lut_table(lut_values, mult: float, clip_min: float, clip_max: float):
    """Host-side decision table for an integer codebook (numpy float32 [K+1, 2]: see include/mctq_hip.h;
    the second word of an entry is a packed half2 bit pattern), or None when the
    codebook / clip range does not qualify (the literal kernels are used then)."""
    import numpy as np
    lib = load()
    k = lib.mctq_lut_table_entries(clip_min, clip_max)
    if k < 0:
        return None
    lut = np.ascontiguousarray(np.asarray(lut_values, dtype=np.float32).reshape(-1))
    table = np.zeros((k + 1, 2), dtype=np.float32)
    rc = lib.mctq_lut_build_table(lut.ctypes.data, lut.size, mult, clip_min, clip_max, table.ctypes.data)
    if rc != 0:
        return None
    return table


# ------------------------------------------------------------------------------------------
# optional roctx ranges (MCTQ_ROCTX=1): one named range per launch, visible in rocprofv3 --marker-trace
# ------------------------------------------------------------------------------------------
_roctx = None


def _roctx_lib():
    global _roctx, TRACE
    if _roctx is None:
        for name in ("librocprofiler-sdk-roctx.so", "libroctx64.so"):
            try:
                lib_ = ctypes.CDLL(name)
                lib_.roctxRangePushA.argtypes = [ctypes.c_char_p]
                lib_.roctxRangePushA.restype = ctypes.c_int
                lib_.roctxRangePop.restype = ctypes.c_int
                _roctx = lib_
                break
            except (OSError, AttributeError):
                continue
        if _roctx is None:
            TRACE = False
            _roctx = False
    return _roctx


class trace_range:
    """``with trace_range("mctq.fq_per_channel")`` -- a roctx range when MCTQ_ROCTX=1, free otherwise."""
    __slots__ = ("name", "on")

    def __init__(self, name: str):
        self.name = name
        self.on = False

    def __enter__(self):
        if TRACE:
            lib_ = _roctx_lib()
            if lib_:
                lib_.roctxRangePushA(self.name.encode())
                self.on = True
        return self

    def __exit__(self, *exc):
        if self.on:
            _roctx.roctxRangePop()
        return False
