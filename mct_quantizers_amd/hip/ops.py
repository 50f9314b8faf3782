"""Tensor-level entry points of the hot path.

Each function stands where the reference calls the tensor runtime:

* ``fq_per_tensor``  ~ ``torch.fake_quantize_per_tensor_affine``
  (reference weights_symmetric_inferable_quantizer.py:147, activation_uniform_inferable_quantizer.py:124, ...)
* ``fq_per_channel`` ~ ``torch.fake_quantize_per_channel_affine`` (weights_symmetric_inferable_quantizer.py:139)
* ``lut_per_tensor`` / ``lut_per_channel`` ~ ``lut_quantizer`` (pytorch/quantizer_utils.py:95-139)

Routing:
  * GPU (HIP) float32 / float16 / bfloat16 / float64 tensor -> the gfx950 kernels through the C ABI, on torch's
    current stream.  No fallback: a missing library raises.  Two interchangeable bindings of the SAME extern "C"
    entry points: the compiled one (csrc/binding/mctq_torch.cpp: checks, allocation, stream lookup and launch in
    one CPython call -- small activations are launch-bound) and ctypes (hip/native.py).
  * CPU tensor                -> the very ATen ops the reference runs on a CPU tensor
    (BASELINE config 1, "torch-cpu plumbing"); not a substitute for the GPU path.
  * torch.jit tracing (TorchScript / ONNX export without enable_custom_impl) -> the ATen operators / torch op
    chain the reference itself records there (a raw kernel launch is invisible to the tracer: the graph would
    contain an uninitialised aten::empty_like).
  * fx Proxy / FakeTensor     -> ``torch.ops.mctq_amd.*`` so tracing records one call_function node.
"""
from __future__ import annotations

from typing import Tuple

import os

import torch

from mct_quantizers_amd.hip import native

_LIBNAME = "mctq_amd"


# ------------------------------------------------------------------------------------------
# layout: view a tensor's storage as [outer][channels][inner]
# ------------------------------------------------------------------------------------------

def _is_dense(x: torch.Tensor) -> bool:
    """True if x occupies numel contiguous elements in some dimension order (no gaps, no overlap)."""
    if x.is_contiguous():
        return True
    dims = [(st, sz) for st, sz in zip(x.stride(), x.shape) if sz != 1]
    dims.sort()
    expect = 1
    for st, sz in dims:
        if st != expect:
            return False
        expect *= sz
    return True


def _dense_input(x: torch.Tensor) -> torch.Tensor:
    """A non-overlapping dense tensor the kernels can walk in storage order.  Gapped / overlapping views are compacted
    the way ATen lays out the OUTPUT of its fake-quant operators for them -- ``empty_like(x)`` with the preserve-format
    rule: dimensions keep their stride order (a gapped channels-last view stays channels-last) -- so the result, which
    takes this tensor's strides, has the strides the reference's result has."""
    return x if _is_dense(x) else torch.empty_like(x).copy_(x)


def _channel_view(x: torch.Tensor, axis: int) -> Tuple[int, int, int]:
    """(outer, channels, inner) of a dense tensor in storage order for logical dimension ``axis``."""
    n = x.numel()
    c = x.shape[axis]
    if x.is_contiguous():
        inner = 1
        for s in x.shape[axis + 1:]:
            inner *= s
    else:
        inner = x.stride(axis) if c > 1 else 1
    outer = n // (c * inner) if c * inner else 0
    return outer, c, inner


# What ONE launch of the non-affine per-channel entry points takes when the rows are short (include/mctq_hip.h, "Size limit"):
# larger tensors are cut into row blocks below it, one launch each.  A module constant so that tests can lower it.
_SPLIT_ELEMS = (1 << 32) - 8192


def _split_rows(x: torch.Tensor, axis: int, params, out_dtype, call, limit: int = None) -> torch.Tensor:
    """``call`` applied block by block to a DENSE tensor too large for one launch.  x is walked in storage order as
    [outer][channels][inner]; a block is a run of whole outer slices, or -- when one slice alone exceeds ``limit`` elements -- a
    run of whole channel rows of one slice, with the per-channel ``params`` (1-D tensors of ``channels`` entries) sliced
    alike.  ``call(x_block [k, ch, inner] contiguous, params_block, 1)`` returns that block's result (same dense order).
    Returns the result with x's sizes and strides."""
    limit = _SPLIT_ELEMS if limit is None else limit
    n = x.numel()
    outer, c, inner = _channel_view(x, axis)
    flat = torch.as_strided(x, (n,), (1,), x.storage_offset())
    y = torch.empty(n, dtype=out_dtype, device=x.device)
    per_slice = c * inner
    if per_slice <= limit:
        k = max(1, limit // per_slice)
        for o0 in range(0, outer, k):
            o1 = min(outer, o0 + k)
            piece = flat[o0 * per_slice:o1 * per_slice].view(o1 - o0, c, inner)
            y[o0 * per_slice:o1 * per_slice] = call(piece, tuple(params), 1).reshape(-1)
    else:
        if inner > limit:
            raise NotImplementedError(f"a single channel row of {inner} elements exceeds what one launch takes ({limit})")
        rows = max(1, limit // inner)
        for o in range(outer):
            for ch0 in range(0, c, rows):
                ch1 = min(c, ch0 + rows)
                a, b = (o * c + ch0) * inner, (o * c + ch1) * inner
                piece = flat[a:b].view(1, ch1 - ch0, inner)
                y[a:b] = call(piece, tuple(p[ch0:ch1] for p in params), 1).reshape(-1)
    return torch.as_strided(y, x.shape, x.stride())


_raw_stream = torch._C._cuda_getCurrentRawStream      # (device index) -> hipStream_t as int
_current_device = torch._C._cuda_getDevice


def _stream(x: torch.Tensor) -> int:
    return _raw_stream(x.get_device())


class _on_device:
    """Make x's device current for the launch (only entered when it is not already current)."""
    __slots__ = ("idx", "prev")

    def __init__(self, idx):
        self.idx = idx
        self.prev = -1

    def __enter__(self):
        self.prev = _current_device()
        torch.cuda.set_device(self.idx)

    def __exit__(self, *exc):
        torch.cuda.set_device(self.prev)
        return False


def _launch(fn, *args):
    """Call a C-ABI launch function; with MCTQ_ROCTX=1 the launch is wrapped in a roctx range named after it."""
    if native.TRACE:
        with native.trace_range(fn.__name__):
            return fn(*args)
    return fn(*args)


class _noop:
    __slots__ = ()

    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NOOP = _noop()


def _maybe_on_device(x):
    idx = x.get_device()
    return _NOOP if idx == _current_device() else _on_device(idx)


_DTYPES = {torch.float32: native.DT_F32, torch.float16: native.DT_F16, torch.bfloat16: native.DT_BF16,
           torch.float64: native.DT_F64}


def _dtype_code(x: torch.Tensor, what: str) -> int:
    code = _DTYPES.get(x.dtype)
    if code is None:
        raise NotImplementedError(f"{what}: the gfx950 kernels take float32, float16, bfloat16 or float64 tensors, "
                                  f"got {x.dtype}")
    return code


def _param_on(x: torch.Tensor, t: torch.Tensor, name: str, dtype) -> torch.Tensor:
    """A parameter vector as the kernels need it: right dtype, contiguous, ON x's DEVICE (the kernel would otherwise
    dereference another GPU's pointer; ATen raises a device-mismatch error here as well)."""
    if t.dtype != dtype:
        raise RuntimeError(f"{name} must be {dtype}, got {t.dtype}")
    if t.device != x.device:
        raise RuntimeError(f"Expected all tensors to be on the same device, but {name} is on {t.device} and the "
                           f"input on {x.device}")
    return t if t.is_contiguous() else t.contiguous()


# compiled binding (None: ctypes only).  Resolved at the first GPU call.
_FAST = None
_FAST_READY = False


def _fast_mod():
    global _FAST, _FAST_READY
    if not _FAST_READY:
        _FAST = native.fast() if torch.cuda.is_available() else None
        _FAST_READY = True
    return _FAST


# ------------------------------------------------------------------------------------------
# GPU launches
# ------------------------------------------------------------------------------------------

def _hip_fq_per_tensor(x, scale: float, zero_point: int, qmin: int, qmax: int):
    # hot for small activations (launch-bound): keep the Python between the caller and the launch short
    dt = _DTYPES.get(x.dtype)
    if dt is None:
        _dtype_code(x, "fq_per_tensor")
    if not qmin <= zero_point <= qmax:                  # ATen's checks and messages
        if qmin > qmax:
            raise RuntimeError("`quant_min` should be less than or         equal to `quant_max`.")
        raise RuntimeError("`zero_point` must be between `quant_min` and `quant_max`.")
    lib = native.load()
    if not x.is_contiguous():
        x = _dense_input(x)
    y = torch.empty_like(x)
    idx = x.get_device()
    if idx == _current_device():
        rc = _launch(lib.mctq_fq_per_tensor, x.data_ptr(), y.data_ptr(), x.numel(), dt, scale, zero_point, qmin, qmax,
                                    _raw_stream(idx))
    else:
        with _on_device(idx):
            rc = _launch(lib.mctq_fq_per_tensor, x.data_ptr(), y.data_ptr(), x.numel(), dt, scale, zero_point, qmin, qmax,
                                        _raw_stream(idx))
    if rc:
        native.check(rc, "mctq_fq_per_tensor")
    return y


def _check_axis(x, n_params: int, axis: int):
    d = x.dim()
    if axis >= d or axis < -d:                    # ATen's own errors, in its order (fake_quantize_per_channel_affine)
        raise IndexError(f"Dimension out of range (expected to be in range of [{-max(d, 1)}, {max(d, 1) - 1}], but got {axis})")
    if axis < 0:
        raise RuntimeError("`axis` must be between 0 and number of dimensions of input")
    if n_params != x.shape[axis]:
        raise RuntimeError("dimensions of scale and zero-point are not consistent with input tensor")


def _hip_fq_per_channel(x, scales, zero_points, axis: int, qmin: int, qmax: int, zero_zps: bool = False):
    """``zero_zps``: the caller knows every zero point is 0 (symmetric quantizers); the table is not read."""
    dt = _DTYPES.get(x.dtype)
    if dt is None:
        _dtype_code(x, "fq_per_channel")
    if not 0 <= axis < x.dim() or scales.numel() != x.shape[axis]:
        _check_axis(x, scales.numel(), axis)
    scales = _param_on(x, scales, "scales", torch.float32)
    zero_points = _param_on(x, zero_points, "zero_points", torch.int32)
    lib = native.load()
    if not x.is_contiguous():
        x = _dense_input(x)
    y = torch.empty_like(x)
    outer, c, inner = _channel_view(x, axis)
    idx = x.get_device()
    with (_NOOP if idx == _current_device() else _on_device(idx)):
        rc = _launch(lib.mctq_fq_per_channel, x.data_ptr(), y.data_ptr(), outer, c, inner, dt, scales.data_ptr(),
                                     None if zero_zps else zero_points.data_ptr(), qmin, qmax, _raw_stream(idx))
    if rc:
        native.check(rc, "mctq_fq_per_channel")
    return y


def _hip_fq_per_tensor_tqp(x, scale, zero_point, qmin: int, qmax: int):
    """Tensor-qparams overload: ``scale`` float32[1] and ``zero_point`` int32[1] stay on the device (no .item())."""
    dt = _dtype_code(x, "fq_per_tensor_tqp")
    if scale.numel() < 1 or zero_point.numel() < 1:      # (ATen reads element 0 of longer tensors; so does the kernel)
        raise RuntimeError("fq_per_tensor_tqp: scale and zero_point must have at least one element")
    scale, zero_point = scale.reshape(-1)[:1], zero_point.reshape(-1)[:1]
    scale = _param_on(x, scale, "scale", torch.float32)
    zero_point = _param_on(x, zero_point, "zero_point", torch.int32)
    lib = native.load()
    if not x.is_contiguous():
        x = _dense_input(x)
    y = torch.empty_like(x)
    with _maybe_on_device(x):
        rc = _launch(lib.mctq_fq_per_tensor_tqp, x.data_ptr(), y.data_ptr(), x.numel(), dt, scale.data_ptr(),
                     zero_point.data_ptr(), qmin, qmax, _stream(x))
    if rc:
        native.check(rc, "mctq_fq_per_tensor_tqp")
    return y


def _hip_fq_batched(items):
    """ctypes route of fq_batched (see below): items = [(x, scales, zero_points | None, axis | None, qmin, qmax)]."""
    lib = native.load()
    n = len(items)
    arr = (native.FqItem * n)()
    outs, keep = [], []
    dev = None
    for k, (x, scales, zps, axis, qmin, qmax) in enumerate(items):
        dt = _dtype_code(x, "fq_batched")
        if dev is None:
            dev = x.device
        elif x.device != dev:
            raise RuntimeError("fq_batched: all tensors of one call must be on the same device")
        if not x.is_contiguous():
            x = _dense_input(x)
        scales = _param_on(x, scales, "scales", torch.float32)
        if zps is not None:
            zps = _param_on(x, zps, "zero_points", torch.int32)
        elif axis is None and dt == native.DT_F64:
            zps = torch.zeros(1, dtype=torch.int32, device=x.device)
        if axis is None:
            outer, c, inner = (1, 1, x.numel()) if x.numel() else (0, 1, 0)
            if scales.numel() != 1:
                raise RuntimeError("fq_batched: a per-tensor item takes 1-element scales / zero_points")
        else:
            _check_axis(x, scales.numel(), axis)
            outer, c, inner = _channel_view(x, axis)
        y = torch.empty_like(x)
        it = arr[k]
        it.x, it.y = x.data_ptr(), y.data_ptr()
        it.outer, it.channels, it.inner = outer, c, inner
        it.scales = scales.data_ptr()
        it.zero_points = zps.data_ptr() if zps is not None else None
        it.quant_min, it.quant_max, it.dtype = qmin, qmax, dt
        it.flags = native.FQ_ITEM_PER_TENSOR if axis is None else 0
        outs.append(y)
        keep.append((x, scales, zps))
    if n:
        with _maybe_on_device(outs[0]):
            rc = _launch(lib.mctq_fq_batched, arr, n, _stream(outs[0]))
        if rc:
            native.check(rc, "mctq_fq_batched")
    return outs


def _lut_result(x, y):
    """The reference's LUT op chain ends in an index gather (quantizer_utils.py:135-137), whose result is a CONTIGUOUS
    tensor whatever the input's memory layout -- unlike ATen's fake-quant operators, which keep the input's strides.
    The kernels work in storage order (same strides as the input); a permuted input pays one layout copy here to hand
    back what the reference hands back."""
    return y if x.is_contiguous() else y.contiguous()


def _hip_lut_per_tensor(x, lut, thr_div: float, thr_mul: float, mult: float, cmin: float, cmax: float, table=None,
                        step_round: int = 0, thr_div64: float = None, steps=None):
    """LUT quantizer, one threshold.  Output is float32 whatever x's type (the reference's chain promotes).
    ``thr_div64``: for float64 tensors whose divisor is a Python float (the activation quantizer) the divisor
    stays a double."""
    if not x.is_floating_point() and not x.is_complex():
        x = x.to(torch.float32)          # integer / bool tensors: the chain's first op, a true division, promotes to float32
    dt = _dtype_code(x, "lut_per_tensor")
    if dt != native.DT_F64 and table is not None and native.TRACE is False:
        f = _FAST if _FAST_READY else _fast_mod()
        if f is not None:
            y = f.lutt_per_tensor(x, table, step_round, thr_div, thr_mul, mult, cmin, cmax)
            if y is not NotImplemented:
                return y if x.is_contiguous() else y.contiguous()      # see _lut_result
    lib = native.load()
    x_in = x
    x = _dense_input(x)
    y = torch.empty_like(x, dtype=torch.float32)
    lut = _param_on(x, lut, "lut_values", torch.float32)
    with _maybe_on_device(x):
        s64 = _op_steps64(lut, mult, cmin, cmax) if dt == native.DT_F64 else None
        if s64 is not None:
            # float64 tensor, integer codebook: the threshold list evaluated in double (the divisor is the activation
            # quantizer's double, or a weights quantizer's float32 sum widened -- the same number either way)
            rc = _launch(lib.mctq_luts_per_tensor_f64, x.data_ptr(), y.data_ptr(), x.numel(),
                         thr_div64 if thr_div64 is not None else thr_div, thr_mul, s64[0].data_ptr(), s64[1], mult, cmin,
                         cmax, _stream(x))
        elif dt == native.DT_F64:
            if thr_div64 is not None:
                rc = _launch(lib.mctq_lut_per_tensor_f64, x.data_ptr(), y.data_ptr(), x.numel(), thr_div64, thr_mul,
                             lut.data_ptr(), lut.numel(), mult, cmin, cmax, _stream(x))
            else:
                rc = _launch(lib.mctq_lut_per_tensor, x.data_ptr(), y.data_ptr(), x.numel(), dt, 0, thr_div, thr_mul,
                             lut.data_ptr(), lut.numel(), mult, cmin, cmax, _stream(x))
        elif table is not None:
            table = _param_on(x, table, "table", torch.float32)
            rc = _launch(lib.mctq_lutt_per_tensor, x.data_ptr(), y.data_ptr(), x.numel(), dt, step_round, thr_div, thr_mul,
                                          table.data_ptr(), table.shape[0] - 1, mult, cmin, cmax, _stream(x))
        elif steps is not None:
            # integer codebook too wide for the table: sorted threshold list, binary search in LDS
            steps = _param_on(x, steps, "steps", torch.float32)
            rc = _launch(lib.mctq_luts_per_tensor, x.data_ptr(), y.data_ptr(), x.numel(), dt, step_round, thr_div, thr_mul,
                         steps.data_ptr(), steps.numel(), mult, cmin, cmax, _stream(x))
        else:
            # literal first-minimum scan (non-integer codebooks, wide bit widths): every storage type, incl. the
            # per-step half-precision roundings of a half activation (step_round)
            rc = _launch(lib.mctq_lut_per_tensor, x.data_ptr(), y.data_ptr(), x.numel(), dt, step_round, thr_div, thr_mul,
                                         lut.data_ptr(), lut.numel(), mult, cmin, cmax, _stream(x))
    if rc:
        native.check(rc, "mctq_lut_per_tensor")
    return _lut_result(x_in, y)


def _hip_lut_per_channel(x, lut, thresholds, eps: float, axis: int, mult: float, cmin: float, cmax: float,
                         table=None, steps=None):
    if not x.is_floating_point() and not x.is_complex():
        x = x.to(torch.float32)          # as in _hip_lut_per_tensor
    dt = _dtype_code(x, "lut_per_channel")
    _check_axis(x, thresholds.numel(), axis)
    if x.numel() > _SPLIT_ELEMS and thresholds.numel() > 1:            # more than one launch of the short-row kernels takes
        y = _split_rows(_dense_input(x), axis, (thresholds,), torch.float32,
                        lambda xp, ps, ax: _hip_lut_per_channel(xp, lut, ps[0], eps, ax, mult, cmin, cmax, table, steps))
        return _lut_result(x, y)
    if dt != native.DT_F64 and table is not None and native.TRACE is False:
        f = _FAST if _FAST_READY else _fast_mod()
        if f is not None:
            y = f.lutt_per_channel(x, thresholds, eps, table, axis, mult, cmin, cmax)
            if y is not NotImplemented:
                return y if x.is_contiguous() else y.contiguous()      # see _lut_result
    lib = native.load()
    x_in = x
    x = _dense_input(x)
    y = torch.empty_like(x, dtype=torch.float32)
    outer, c, inner = _channel_view(x, axis)
    thresholds = _param_on(x, thresholds, "thresholds", torch.float32)
    lut = _param_on(x, lut, "lut_values", torch.float32)
    with _maybe_on_device(x):
        s64 = _op_steps64(lut, mult, cmin, cmax) if dt == native.DT_F64 else None
        if s64 is not None:
            rc = _launch(lib.mctq_luts_per_channel_f64, x.data_ptr(), y.data_ptr(), outer, c, inner, thresholds.data_ptr(),
                         eps, s64[0].data_ptr(), s64[1], mult, cmin, cmax, _stream(x))
        elif table is not None and dt != native.DT_F64:
            table = _param_on(x, table, "table", torch.float32)
            rc = _launch(lib.mctq_lutt_per_channel, x.data_ptr(), y.data_ptr(), outer, c, inner, dt, thresholds.data_ptr(),
                                           eps, table.data_ptr(), table.shape[0] - 1, mult, cmin, cmax, _stream(x))
        elif steps is not None and dt != native.DT_F64:
            steps = _param_on(x, steps, "steps", torch.float32)
            rc = _launch(lib.mctq_luts_per_channel, x.data_ptr(), y.data_ptr(), outer, c, inner, dt, thresholds.data_ptr(),
                         eps, steps.data_ptr(), steps.numel(), mult, cmin, cmax, _stream(x))
        else:
            rc = _launch(lib.mctq_lut_per_channel, x.data_ptr(), y.data_ptr(), outer, c, inner, dt, thresholds.data_ptr(),
                                          eps, lut.data_ptr(), lut.numel(), mult, cmin, cmax, _stream(x))
    if rc:
        native.check(rc, "mctq_lut_per_channel")
    return _lut_result(x_in, y)


def _hip_grid_per_tensor(x, lo: float, hi: float, step: float, shifted: bool):
    """Export-time arithmetic (include/mctq_hip.h: mctq_grid_per_tensor_f32), float32 only."""
    if x.dtype != torch.float32:
        raise NotImplementedError(f"export-time quantizer arithmetic on the GPU takes float32 tensors, got {x.dtype}")
    lib = native.load()
    x = _dense_input(x)
    y = torch.empty_like(x)
    with _maybe_on_device(x):
        rc = _launch(lib.mctq_grid_per_tensor_f32, x.data_ptr(), y.data_ptr(), x.numel(), lo, hi, step, int(shifted),
                     _stream(x))
    if rc:
        native.check(rc, "mctq_grid_per_tensor_f32")
    return y


def _hip_grid_per_channel(x, los, his, steps, axis: int, shifted: bool):
    if x.dtype != torch.float32:
        raise NotImplementedError(f"export-time quantizer arithmetic on the GPU takes float32 tensors, got {x.dtype}")
    _check_axis(x, steps.numel(), axis)
    lib = native.load()
    x = _dense_input(x)
    if x.numel() > _SPLIT_ELEMS and steps.numel() > 1:
        return _split_rows(x, axis, (los.reshape(-1), his.reshape(-1), steps.reshape(-1)), torch.float32,
                           lambda xp, ps, ax: _hip_grid_per_channel(xp, ps[0], ps[1], ps[2], ax, shifted))
    y = torch.empty_like(x)
    outer, c, inner = _channel_view(x, axis)
    los, his, steps = (t.to(device=x.device, dtype=torch.float32).contiguous() for t in (los, his, steps))
    with _maybe_on_device(x):
        rc = _launch(lib.mctq_grid_per_channel_f32, x.data_ptr(), y.data_ptr(), outer, c, inner, los.data_ptr(),
                     his.data_ptr(), steps.data_ptr(), int(shifted), _stream(x))
    if rc:
        native.check(rc, "mctq_grid_per_channel_f32")
    return y


def _code_dtype(qmin: int, qmax: int):
    if qmin >= 0 and qmax <= 255:
        return torch.uint8, native.CODE_U8
    if qmin >= -128 and qmax <= 127:
        return torch.int8, native.CODE_I8
    raise ValueError(f"clamp domain [{qmin}, {qmax}] does not fit an 8-bit code")


def _packed_shape(x):
    return tuple(x.shape[:-1]) + (x.shape[-1] // 2,) if x.dim() and x.shape[-1] % 2 == 0 and x.is_contiguous() \
        else (x.numel() // 2,)


def pack4(q: torch.Tensor) -> torch.Tensor:
    """Integer codes in [-8, 15] (any integer dtype, storage order) -> two per byte, element 2j in the low nibble."""
    flat = q.reshape(-1).to(torch.int16) & 0xF
    return (flat[0::2] | (flat[1::2] << 4)).to(torch.uint8)


def unpack4(packed: torch.Tensor, signed: bool, shape=None) -> torch.Tensor:
    """Inverse of the 4-bit packing: uint8 [n/2] (any shape) -> int8 codes [n] (or ``shape``)."""
    b = packed.reshape(-1).to(torch.int16)
    both = torch.stack((b & 0xF, (b >> 4) & 0xF), dim=1).reshape(-1)
    if signed:
        both = torch.where(both > 7, both - 16, both)
    both = both.to(torch.int8)
    return both if shape is None else both.reshape(shape)


def fq_codes(x, scales, zero_points, axis, qmin: int, qmax: int, scale0: float = None, zp0: int = None,
             packed4: bool = False):
    """Integer clamp indices of the affine quantizers as int8 / uint8 (extension, not in the reference).

    ``axis`` None = per-tensor (scale0 / zp0 are the host copies of the single scale and zero point).
    ``(codes - zero_point) * scale`` is bit-identical to the fake-quantized tensor.
    ``packed4``: clamp domains within [-8, 7] or [0, 15] leave as two codes per byte (uint8 tensor of half the
    elements: last dimension halved for contiguous tensors; element 2j of the storage order in the low nibble).
    """
    if packed4:
        if not ((qmin >= -8 and qmax <= 7) or (qmin >= 0 and qmax <= 15)):
            raise ValueError(f"clamp domain [{qmin}, {qmax}] does not fit a 4-bit code")
        if x.numel() % 2:
            raise ValueError("4-bit packing needs an even number of elements")
    tdt, code = _code_dtype(qmin, qmax)
    if not x.is_cuda:
        # CPU tensors: the same arithmetic with torch ops (float32 math, as ATen's CPU kernel)
        xf = x.float()
        if axis is None:
            q = torch.round(xf * (torch.tensor(1.0, dtype=torch.float32) / torch.tensor(scale0, dtype=torch.float32))) + zp0
        else:
            shape = [1] * x.dim()
            shape[axis] = -1
            q = torch.round(xf * (1.0 / scales.float().to(x.device)).reshape(shape)) + \
                zero_points.to(x.device).reshape(shape).float()
        q = torch.clamp(torch.nan_to_num(q, nan=float(qmin)), qmin, qmax)
        if packed4:
            if x.is_contiguous() or not _is_dense(x):
                return pack4(q.contiguous()).reshape(_packed_shape(x))
            flat = torch.empty(x.numel(), dtype=torch.int16)          # dense, permuted storage: pack in STORAGE order
            torch.as_strided(flat, x.shape, x.stride()).copy_(q.to(torch.int16))
            return pack4(flat).reshape(_packed_shape(x))
        return q.to(tdt)
    dt = _dtype_code(x, "fq_codes")
    if dt == native.DT_F64:
        raise NotImplementedError("fq_codes: integer codes are produced from float32 / float16 / bfloat16 tensors")
    lib = native.load()
    if not x.is_contiguous():
        x = _dense_input(x)
    if axis is not None and not packed4 and x.numel() > _SPLIT_ELEMS and scales.numel() > 1:
        return _split_rows(x, axis, (scales.reshape(-1), zero_points.reshape(-1)), tdt,
                           lambda xp, ps, ax: fq_codes(xp, ps[0], ps[1], ax, qmin, qmax))
    if packed4:
        code = native.CODE_I4 if qmin < 0 else native.CODE_U4
        y = torch.empty(_packed_shape(x), dtype=torch.uint8, device=x.device)
    else:
        y = torch.empty_like(x, dtype=tdt)
    idx = x.get_device()
    with (_NOOP if idx == _current_device() else _on_device(idx)):
        if axis is None:
            rc = _launch(lib.mctq_fq_codes_per_tensor, x.data_ptr(), y.data_ptr(), x.numel(), dt, code, scale0, zp0, qmin, qmax,
                                              _raw_stream(idx))
        else:
            _check_axis(x, scales.numel(), axis)
            scales = _param_on(x, scales, "scales", torch.float32)
            zero_points = _param_on(x, zero_points, "zero_points", torch.int32)
            outer, c, inner = _channel_view(x, axis)
            rc = _launch(lib.mctq_fq_codes_per_channel, x.data_ptr(), y.data_ptr(), outer, c, inner, dt, code,
                                               scales.data_ptr(), zero_points.data_ptr(), qmin, qmax, _raw_stream(idx))
    if rc:
        native.check(rc, "mctq_fq_codes")
    return y


def fq_codes_nhwc(x, qmin: int, qmax: int, scale: float, zero_point: int):
    """Per-tensor codes of a 4-D activation as a contiguous [N, H, W, C] int8 / uint8 tensor, whatever x's memory
    format.  NCHW-contiguous GPU tensors take one fused quantize-and-transpose pass (mctq_fq_codes_nchw_to_nhwc);
    channels-last ones are quantized in place (their storage order already is NHWC)."""
    if x.dim() != 4:
        raise ValueError("fq_codes_nhwc takes [N, C, H, W] tensors")
    b, c, h, w = x.shape
    if x.is_cuda and x.is_contiguous() and not x.is_contiguous(memory_format=torch.channels_last):
        tdt, code = _code_dtype(qmin, qmax)
        dt = _dtype_code(x, "fq_codes_nhwc")
        if dt == native.DT_F64:
            raise NotImplementedError("fq_codes_nhwc: float32 / float16 / bfloat16 tensors only")
        lib = native.load()
        y = torch.empty((b, h, w, c), dtype=tdt, device=x.device)
        with _maybe_on_device(x):
            rc = _launch(lib.mctq_fq_codes_nchw_to_nhwc, x.data_ptr(), y.data_ptr(), b, c, h * w, dt, code, float(scale),
                         int(zero_point), qmin, qmax, _stream(x))
        if rc:
            native.check(rc, "mctq_fq_codes_nchw_to_nhwc")
        return y
    codes = fq_codes(x, None, None, None, qmin, qmax, scale, zero_point)
    return codes.permute(0, 2, 3, 1).contiguous()          # a no-op view + check for channels-last storage


def make_lut_table(lut_values, mult: float, cmin: float, cmax: float, device):
    """Device copy of the codebook's decision table (see include/mctq_hip.h), or None.

    Built once per quantizer at construction; needs the native library only when a GPU is the
    working device."""
    if torch.device(device).type != "cuda":
        return None
    table = native.build_lut_table(lut_values, mult, cmin, cmax)
    if table is None:
        return None
    return torch.from_numpy(table).to(device)


def make_lut_steps(lut_values, mult: float, cmin: float, cmax: float, device):
    """Device copy of an integer codebook's sorted threshold list (include/mctq_hip.h: mctq_lut_build_steps), or None.
    The quantizers ask for it only when the decision table does not apply (lut_values_bitwidth > 10)."""
    if torch.device(device).type != "cuda":
        return None
    steps = native.build_lut_steps(lut_values, mult, cmin, cmax)
    if steps is None:
        return None
    return torch.from_numpy(steps).to(device)


# ------------------------------------------------------------------------------------------
# CPU tensors: the ATen ops / op chain the reference itself executes on a CPU tensor
# ------------------------------------------------------------------------------------------

def _cpu_fq_per_tensor(x, scale, zero_point, qmin, qmax):
    return torch.fake_quantize_per_tensor_affine(x, scale, zero_point, qmin, qmax)


def _cpu_fq_per_channel(x, scales, zero_points, axis, qmin, qmax):
    return torch.fake_quantize_per_channel_affine(x, scales, zero_points, axis, qmin, qmax)


def _cpu_lut(x, lut, thr_div, thr_mul, mult, cmin, cmax):
    # quantizer_utils.py:126-137 on CPU tensors: divide, scale, clip, first-min argmin, gather, rescale
    t = torch.clip((x / thr_div) * mult, min=cmin, max=cmax).unsqueeze(-1)
    idx = torch.argmin(torch.abs(t - lut.reshape([1] * (t.dim() - 1) + [-1])), dim=-1)
    return (lut.flatten()[idx] / mult) * thr_mul


def _cpu_lut_per_tensor(x, lut, thr_div, thr_mul, mult, cmin, cmax, step_round: int = 0):
    # step_round == 0: the threshold is a float32 TENSOR in the reference (weights), so a half-precision
    # input is promoted to float32 by the first division; otherwise (activation, Python-float threshold)
    # the chain stays in the input's type until the float32 codebook enters.
    # (a float64 tensor stays float64 either way: the double quotient, clip and distances of ATen's promotion.)
    if step_round == 0 and x.dtype in (torch.float16, torch.bfloat16):
        x = x.float()
    return _cpu_lut(x, lut, thr_div, thr_mul, mult, cmin, cmax)


def _cpu_lut_per_channel(x, lut, thresholds, eps, axis, mult, cmin, cmax):
    shape = [1] * x.dim()
    shape[axis] = -1
    thr = thresholds.reshape(shape)
    return _cpu_lut(x, lut, thr + eps, thr, mult, cmin, cmax)


def _cpu_grid(x, lo, hi, step, shifted: bool):
    # the reference's export-time chains on a CPU tensor (weights_symmetric...py:67-68, activation_uniform...py:60-64);
    # lo / hi / step: Python floats or broadcastable float32 tensors
    c = torch.where(x < lo, lo, x)
    c = torch.where(x > hi, hi, c)
    if shifted:
        return step * torch.round((c - lo) / step) + lo
    return torch.round(c / step) * step


# ------------------------------------------------------------------------------------------
# torch.library registration (fx tracing / FakeTensor / torch.compile see one opaque op)
# ------------------------------------------------------------------------------------------

_lib_def = torch.library.Library(_LIBNAME, "DEF")
_lib_def.define("fq_per_tensor(Tensor x, float scale, int zero_point, int quant_min, int quant_max) -> Tensor")
_lib_def.define("fq_per_tensor_tqp(Tensor x, Tensor scale, Tensor zero_point, int quant_min, int quant_max) -> Tensor")
_lib_def.define("fq_per_channel(Tensor x, Tensor scales, Tensor zero_points, int axis, int quant_min, "
                "int quant_max) -> Tensor")
_lib_def.define("lut_per_tensor(Tensor x, Tensor lut, float thr_div, float thr_mul, float mult, float clip_min, "
                "float clip_max, int step_round=0) -> Tensor")
_lib_def.define("lut_per_channel(Tensor x, Tensor lut, Tensor thresholds, float eps, int axis, float mult, "
                "float clip_min, float clip_max) -> Tensor")

# decision tables for codebooks that reach the ops without their quantizer (fx graphs): built once per codebook TENSOR
# OBJECT (weak reference + version counter: an address can be reused by another tensor, an object cannot)
_op_tables = {}


def _op_table(lut, mult, cmin, cmax):
    """(decision table | None, threshold list | None) of a codebook tensor."""
    import weakref
    key = (id(lut), mult, cmin, cmax)
    hit = _op_tables.get(key)
    if hit is not None and hit[0]() is lut and hit[1] == lut._version:
        return hit[2]
    if len(_op_tables) > 256:
        _op_tables.clear()
    lut_np = lut.detach().cpu().numpy()
    table = make_lut_table(lut_np, mult, cmin, cmax, lut.device)
    books = (table, None if table is not None else make_lut_steps(lut_np, mult, cmin, cmax, lut.device))
    try:
        _op_tables[key] = (weakref.ref(lut), lut._version, books)
    except TypeError:                                   # not weak-referenceable: do not cache
        pass
    return books


_op_steps64_cache = {}


def _op_steps64(lut, mult, cmin, cmax):
    """(device blob, P) of the DOUBLE threshold list of a codebook tensor for float64 inputs, or None (literal double
    scan).  Built once per codebook TENSOR OBJECT and version, like ``_op_table`` (one device -> host read at the first
    float64 call: not inside hipGraph capture)."""
    import weakref
    key = (id(lut), mult, cmin, cmax)
    hit = _op_steps64_cache.get(key)
    if hit is not None and hit[0]() is lut and hit[1] == lut._version:
        return hit[2]
    if len(_op_steps64_cache) > 256:
        _op_steps64_cache.clear()
    built = native.build_lut_steps_f64(lut.detach().cpu().numpy(), mult, cmin, cmax)
    val = None if built is None else (torch.from_numpy(built[0]).to(lut.device), built[1])
    try:
        _op_steps64_cache[key] = (weakref.ref(lut), lut._version, val)
    except TypeError:
        pass
    return val


def _op_hip_lut_per_tensor(x, lut, thr_div, thr_mul, mult, cmin, cmax, step_round=0):
    if x.dtype == torch.float64:
        # the schema's float is a double: the activation quantizer's double divisor arrives intact, and a weights
        # quantizer's float32 divisor is the same number either way
        return _hip_lut_per_tensor(x, lut, thr_div, thr_mul, mult, cmin, cmax, None, 0, thr_div)
    table, steps = _op_table(lut, mult, cmin, cmax)
    return _hip_lut_per_tensor(x, lut, thr_div, thr_mul, mult, cmin, cmax, table, step_round, None, steps)


def _op_hip_lut_per_channel(x, lut, thresholds, eps, axis, mult, cmin, cmax):
    if x.dtype == torch.float64:
        return _hip_lut_per_channel(x, lut, thresholds, eps, axis, mult, cmin, cmax, None)
    table, steps = _op_table(lut, mult, cmin, cmax)
    return _hip_lut_per_channel(x, lut, thresholds, eps, axis, mult, cmin, cmax, table, steps)


def _cpu_fq_per_tensor_tqp(x, scale, zero_point, qmin, qmax):
    return torch.fake_quantize_per_tensor_affine(x, scale, zero_point, qmin, qmax)


for _name, _gpu, _cpu in (("fq_per_tensor", _hip_fq_per_tensor, _cpu_fq_per_tensor),
                          ("fq_per_tensor_tqp", _hip_fq_per_tensor_tqp, _cpu_fq_per_tensor_tqp),
                          ("fq_per_channel", _hip_fq_per_channel, _cpu_fq_per_channel),
                          ("lut_per_tensor", _op_hip_lut_per_tensor, _cpu_lut_per_tensor),
                          ("lut_per_channel", _op_hip_lut_per_channel, _cpu_lut_per_channel)):
    _lib_def.impl(_name, _gpu, "CUDA")
    _lib_def.impl(_name, _cpu, "CPU")
for _name in ("fq_per_tensor", "fq_per_tensor_tqp", "fq_per_channel"):
    _lib_def.impl(_name, (lambda x, *a: torch.empty_like(x)), "Meta")
for _name in ("lut_per_tensor", "lut_per_channel"):          # the LUT chain's result: float32, contiguous (_lut_result)
    _lib_def.impl(_name, (lambda x, *a: torch.empty(x.shape, dtype=torch.float32, device=x.device)), "Meta")


# ---- autograd: what ATen's operators do at the reference's call sites -----------------------------------------
# torch.fake_quantize_per_{tensor,channel}_affine carry a straight-through backward: the incoming gradient where the
# clamp index q = nearbyint(x * (1.0f / scale)) + zero_point lies inside [quant_min, quant_max], zero elsewhere
# (ATen's *_cachemask kernels record that mask in the forward; here it is recomputed from x in the rare backward, with
# plain torch ops -- this is an inference library, the reference switches gradients off before it calls them:
# weights_symmetric_inferable_quantizer.py:138,146, activation_symmetric_inferable_quantizer.py:112).
# The scale / zero-point tensors get no gradient (nor do they in ATen's tensor-qparams and per-channel operators).
# The LUT chain ends in argmin + gather (quantizer_utils.py:131-137): its result does not depend on x differentiably.

def _ste_mask(x, inv_scale, zero_point, qmin: int, qmax: int):
    q = torch.round(x.detach().float() * inv_scale) + zero_point          # round half to even, as nearbyint
    return (q >= qmin) & (q <= qmax)


def _f32_inverse(scale):
    s = scale if isinstance(scale, torch.Tensor) else torch.tensor(float(scale), dtype=torch.float64)
    return torch.tensor(1.0, dtype=torch.float32, device=s.device) / s.detach().to(torch.float32)    # float32 1.0f / scale


def _ctx_per_tensor(ctx, inputs, output):
    x, scale, zero_point, qmin, qmax = inputs
    ctx.save_for_backward(x)
    ctx.q = (scale, zero_point, qmin, qmax)


def _bwd_per_tensor(ctx, grad):
    (x,) = ctx.saved_tensors
    scale, zero_point, qmin, qmax = ctx.q
    mask = _ste_mask(x, _f32_inverse(scale).to(x.device), zero_point, qmin, qmax)
    return grad * mask, None, None, None, None


def _ctx_per_tensor_tqp(ctx, inputs, output):
    x, scale, zero_point, qmin, qmax = inputs
    ctx.save_for_backward(x, scale, zero_point)
    ctx.q = (qmin, qmax)


def _bwd_per_tensor_tqp(ctx, grad):
    x, scale, zero_point = ctx.saved_tensors
    qmin, qmax = ctx.q
    mask = _ste_mask(x, _f32_inverse(scale.reshape(-1)[:1]), zero_point.reshape(-1)[:1].to(torch.float32), qmin, qmax)
    return grad * mask, None, None, None, None


def _ctx_per_channel(ctx, inputs, output):
    x, scales, zero_points, axis, qmin, qmax = inputs
    ctx.save_for_backward(x, scales, zero_points)
    ctx.q = (axis, qmin, qmax)


def _bwd_per_channel(ctx, grad):
    x, scales, zero_points = ctx.saved_tensors
    axis, qmin, qmax = ctx.q
    shape = [1] * x.dim()
    shape[axis] = -1
    mask = _ste_mask(x, _f32_inverse(scales).reshape(shape), zero_points.to(torch.float32).reshape(shape), qmin, qmax)
    return grad * mask, None, None, None, None, None


def _ctx_nondiff(ctx, inputs, output):
    ctx.mark_non_differentiable(output)


torch.library.register_autograd(f"{_LIBNAME}::fq_per_tensor", _bwd_per_tensor, setup_context=_ctx_per_tensor, lib=_lib_def)
torch.library.register_autograd(f"{_LIBNAME}::fq_per_tensor_tqp", _bwd_per_tensor_tqp, setup_context=_ctx_per_tensor_tqp,
                                lib=_lib_def)
torch.library.register_autograd(f"{_LIBNAME}::fq_per_channel", _bwd_per_channel, setup_context=_ctx_per_channel, lib=_lib_def)
torch.library.register_autograd(f"{_LIBNAME}::lut_per_tensor", lambda ctx, grad: (None,) * 8, setup_context=_ctx_nondiff,
                                lib=_lib_def)
torch.library.register_autograd(f"{_LIBNAME}::lut_per_channel", lambda ctx, grad: (None,) * 8, setup_context=_ctx_nondiff,
                                lib=_lib_def)


def _is_real(x) -> bool:
    return type(x) is torch.Tensor or type(x) is torch.nn.Parameter


def _cpu_route_allowed():
    """CPU tensors run the ATen operators the reference runs on them (BASELINE config 1).  Deployments that
    must never leave the GPU set MCTQ_REQUIRE_HIP=1: a CPU tensor then raises instead."""
    if os.environ.get("MCTQ_REQUIRE_HIP", "0") not in ("", "0"):
        raise RuntimeError("MCTQ_REQUIRE_HIP is set: refusing to quantize a CPU tensor outside the HIP kernels")


def _tracing() -> bool:
    """torch.jit.trace / torch.onnx.export in progress: emit what the reference emits (ATen nodes), never a raw launch."""
    return torch._C._get_tracing_state() is not None


_compiling = torch.compiler.is_compiling      # dynamo / export in progress: the graph must see torch.ops.mctq_amd.*


def _wide(qmin, qmax) -> bool:
    """Clamp domains beyond 2^24 do not fit the kernels' float32 bounds: ATen's operator (the reference's own call) runs."""
    return max(abs(int(qmin)), abs(int(qmax))) > (1 << 24)


def fq_per_tensor(x, scale: float, zero_point: int, qmin: int, qmax: int):
    if _wide(qmin, qmax) and _is_real(x):
        return torch.fake_quantize_per_tensor_affine(x, scale, zero_point, qmin, qmax)
    if _compiling():
        return torch.ops.mctq_amd.fq_per_tensor(x, scale, zero_point, qmin, qmax)
    f = _FAST if _FAST_READY else _fast_mod()
    if f is not None:
        y = f.fq_per_tensor(x, scale, zero_point, qmin, qmax)
        if y is not NotImplemented:
            return y
    if _is_real(x):
        if _tracing():
            return torch.fake_quantize_per_tensor_affine(x, scale, zero_point, qmin, qmax)
        if x.is_cuda:
            return _hip_fq_per_tensor(x, scale, zero_point, qmin, qmax)
        if x.device.type == "cpu":
            _cpu_route_allowed()
            return _cpu_fq_per_tensor(x, scale, zero_point, qmin, qmax)
    return torch.ops.mctq_amd.fq_per_tensor(x, scale, zero_point, qmin, qmax)


def fq_per_tensor_tqp(x, scale, zero_point, qmin: int, qmax: int):
    """``torch.fake_quantize_per_tensor_affine(x, scale_tensor, zero_point_tensor, qmin, qmax)``: the parameters are
    1-element tensors (float32 / int32) that the kernel reads on the device."""
    if _wide(qmin, qmax) and _is_real(x):
        return torch.fake_quantize_per_tensor_affine(x, scale, zero_point, qmin, qmax)
    if _compiling():
        return torch.ops.mctq_amd.fq_per_tensor_tqp(x, scale, zero_point, qmin, qmax)
    f = _FAST if _FAST_READY else _fast_mod()
    if f is not None:
        y = f.fq_per_tensor_tqp(x, scale, zero_point, qmin, qmax)
        if y is not NotImplemented:
            return y
    if _is_real(x):
        if _tracing():
            return torch.fake_quantize_per_tensor_affine(x, scale, zero_point, qmin, qmax)
        if x.is_cuda:
            return _hip_fq_per_tensor_tqp(x, scale, zero_point, qmin, qmax)
        if x.device.type == "cpu":
            _cpu_route_allowed()
            return _cpu_fq_per_tensor_tqp(x, scale, zero_point, qmin, qmax)
    return torch.ops.mctq_amd.fq_per_tensor_tqp(x, scale, zero_point, qmin, qmax)


def fq_per_channel(x, scales, zero_points, axis: int, qmin: int, qmax: int, zero_zps: bool = False):
    if _wide(qmin, qmax) and _is_real(x):
        return torch.fake_quantize_per_channel_affine(x, scales, zero_points, axis, qmin, qmax)
    if _compiling():
        return torch.ops.mctq_amd.fq_per_channel(x, scales, zero_points, axis, qmin, qmax)
    f = _FAST if _FAST_READY else _fast_mod()
    if f is not None:
        y = f.fq_per_channel(x, scales, None if zero_zps else zero_points, axis, qmin, qmax)
        if y is not NotImplemented:
            return y
    if _is_real(x):
        if _tracing():
            return torch.fake_quantize_per_channel_affine(x, scales, zero_points, axis, qmin, qmax)
        if x.is_cuda:
            return _hip_fq_per_channel(x, scales, zero_points, axis, qmin, qmax, zero_zps)
        if x.device.type == "cpu":
            _cpu_route_allowed()
            return _cpu_fq_per_channel(x, scales, zero_points, axis, qmin, qmax)
    return torch.ops.mctq_amd.fq_per_channel(x, scales, zero_points, axis, qmin, qmax)


def fq_batched(items):
    """Affine fake-quantization of a LIST of tensors in one launch (per group of 32 tensors of one storage type).

    ``items``: sequence of ``(x, scales, zero_points | None, axis | None, quant_min, quant_max)``; ``axis`` None =
    per tensor with 1-element device ``scales`` / ``zero_points``.  All tensors dense GPU tensors on one device.
    Returns the list of outputs, each bit-identical to the corresponding single call.  Replaces the per-layer
    quantizer calls of PytorchQuantizationWrapper.forward (pytorch/quantize_wrapper.py:228-240) when a whole
    model's weights are re-quantized together (see pytorch/batching.py)."""
    items = list(items)
    f = _FAST if _FAST_READY else _fast_mod()
    if any(_wide(it[4], it[5]) for it in items):
        f = None
    if f is not None:
        ys = f.fq_batched(items)
        if ys is not NotImplemented:
            return ys
    if all(_is_real(it[0]) and it[0].is_cuda and not _wide(it[4], it[5]) for it in items) and not _tracing():
        return _hip_fq_batched(items)
    out = []
    for x, scales, zps, axis, qmin, qmax in items:            # CPU tensors, traced graphs ...: one by one
        if zps is None:
            zps = torch.zeros(scales.numel(), dtype=torch.int32, device=scales.device)
        if axis is None:
            out.append(fq_per_tensor_tqp(x, scales, zps, qmin, qmax))
        else:
            out.append(fq_per_channel(x, scales, zps, axis, qmin, qmax))
    return out


def lut_per_tensor(x, lut, thr_div: float, thr_mul: float, mult: float, cmin: float, cmax: float, table=None,
                   step_round: int = 0, thr_div64: float = None, steps=None):
    if _compiling():
        return torch.ops.mctq_amd.lut_per_tensor(x, lut, thr_div if thr_div64 is None else thr_div64, thr_mul, mult, cmin,
                                                 cmax, max(step_round, 0))
    if _is_real(x):
        if _tracing():                                        # the reference's own op chain is what gets recorded
            return _cpu_lut_per_tensor(x, lut, thr_div if thr_div64 is None else thr_div64, thr_mul, mult, cmin, cmax,
                                       step_round)
        if x.is_cuda:
            return _hip_lut_per_tensor(x, lut, thr_div, thr_mul, mult, cmin, cmax, table, step_round, thr_div64, steps)
        if x.device.type == "cpu":
            _cpu_route_allowed()
            return _cpu_lut_per_tensor(x, lut, thr_div, thr_mul, mult, cmin, cmax, step_round)
    return torch.ops.mctq_amd.lut_per_tensor(x, lut, thr_div, thr_mul, mult, cmin, cmax, max(step_round, 0))


def lut_per_channel(x, lut, thresholds, eps: float, axis: int, mult: float, cmin: float, cmax: float, table=None,
                    steps=None):
    if _compiling():
        return torch.ops.mctq_amd.lut_per_channel(x, lut, thresholds, eps, axis, mult, cmin, cmax)
    if _is_real(x):
        if _tracing():
            return _cpu_lut_per_channel(x, lut, thresholds, eps, axis, mult, cmin, cmax)
        if x.is_cuda:
            return _hip_lut_per_channel(x, lut, thresholds, eps, axis, mult, cmin, cmax, table, steps)
        if x.device.type == "cpu":
            _cpu_route_allowed()
            return _cpu_lut_per_channel(x, lut, thresholds, eps, axis, mult, cmin, cmax)
    return torch.ops.mctq_amd.lut_per_channel(x, lut, thresholds, eps, axis, mult, cmin, cmax)


def grid_per_tensor(x, lo: float, hi: float, step: float, shifted: bool = False):
    """clip -> true division -> round half even -> scale back, one parameter set (export-time arithmetic)."""
    f32 = lambda v: torch.tensor(v, dtype=torch.float64).to(torch.float32)    # noqa: E731  (scalar -> float32, RNE)
    if x.is_cuda:
        if x.dtype is torch.float32:
            return _hip_grid_per_tensor(x, lo, hi, step, shifted)
        # export of a model kept in another storage type (once per export): the reference's op chain on the device
        return _cpu_grid(x, f32(lo).to(x.device), f32(hi).to(x.device), f32(step).to(x.device), shifted)
    _cpu_route_allowed()
    return _cpu_grid(x, f32(lo), f32(hi), f32(step), shifted)


def grid_per_channel(x, los, his, steps, axis: int, shifted: bool = False):
    """Same with float32 parameter vectors along ``axis``."""
    if x.is_cuda and x.dtype is torch.float32:
        return _hip_grid_per_channel(x, los, his, steps, axis, shifted)
    if not x.is_cuda:
        _cpu_route_allowed()
    shape = [1] * x.dim()
    shape[axis] = -1
    los, his, steps = (t.to(x.device).reshape(shape) for t in (los, his, steps))     # non-float32 GPU tensors: see above
    return _cpu_grid(x, los, his, steps, shifted)
