"""CPU oracle for the inferable-quantizer hot path (TEST INFRASTRUCTURE, NOT PRODUCT CODE).

This file restates, in plain numpy float32 arithmetic, what sony/mct_quantizers'
PyTorch inferable quantizers compute on a CPU tensor.  It exists so the HIP
kernels can be checked bit for bit.  Only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import it; the shipped package
(``mct_quantizers_amd``) never does.

Parity pin: PINNED.  Every function below is compared bit-exactly with the
reference imported from ``/root/reference`` (torch 2.10 CPU) by
``tools/gen_golden.py`` and against the committed fixtures under
``tests/golden/`` by ``tests/test_oracle_golden.py``.

Third-party arithmetic: six of the nine quantizers delegate to torch ATen
``fake_quantize_per_tensor_affine`` / ``fake_quantize_per_channel_affine``
(not under /root/reference; the reference does not pin torch,
requirements.txt:1-2; fixtures were produced with torch 2.10.0).  The published
ATen CPU algorithm restated here is

    inv = 1.0f / scale
    q   = clamp(nearbyint(x * inv) + zero_point, quant_min, quant_max)
    y   = (q - zero_point) * scale

Citations ``<file>:<line>`` are relative to /root/reference/mct_quantizers/.
"""
from __future__ import annotations

import numpy as np

F32 = np.float32

EPS = 1e-8                 # common/constants.py:83
LUT_VALUES_BITWIDTH = 8    # common/constants.py:84


# --------------------------------------------------------------------------
# parameter derivation (constructors)
# --------------------------------------------------------------------------

def symmetric_domain(num_bits: int, signed: bool):
    """Integer clamp domain, pytorch/quantizers/base_symmetric_inferable_quantizer.py:53-60."""
    if signed:
        return -2 ** (num_bits - 1), 2 ** (num_bits - 1) - 1
    return 0, 2 ** num_bits - 1


def symmetric_scales_f64(num_bits: int, threshold, signed: bool) -> np.ndarray:
    """float64 scales, base_symmetric_inferable_quantizer.py:56,60."""
    thr = np.asarray(threshold)
    return thr / 2 ** (num_bits - 1) if signed else thr / 2 ** num_bits


def weights_symmetric_params(num_bits: int, threshold):
    """(scales fp32[C], zero_points int32[C], qmin, qmax).

    weights_symmetric_inferable_quantizer.py:114-115: float64 scales are cast to
    float32 by to_torch_tensor (quantizer_utils.py:50-51); zero points are zeros.
    """
    scales = symmetric_scales_f64(num_bits, threshold, True).astype(F32)
    zps = np.zeros(len(threshold), dtype=np.int32)
    qmin, qmax = symmetric_domain(num_bits, True)
    return scales, zps, qmin, qmax


def activation_symmetric_params(num_bits: int, threshold, signed: bool):
    """(scale python float, zero_point 0, qmin, qmax).

    activation_symmetric_inferable_quantizer.py:92-94: the scale stays a Python
    double; ATen casts it to float32 when the kernel is launched.
    """
    scale = float(symmetric_scales_f64(num_bits, threshold, signed)[0])
    qmin, qmax = symmetric_domain(num_bits, signed)
    return scale, 0, qmin, qmax


def fix_range_to_include_zero(range_min, range_max, n_bits: int):
    """float32 restatement of pytorch/quantizer_utils.py:60-92 (torch fp32 ops).

    Note there is no final clamp to <=0 / >=0, unlike the numpy twin in
    common/quant_utils.py:47-48.
    """
    rmin = np.asarray(range_min, dtype=F32)
    rmax = np.asarray(range_max, dtype=F32)
    min_positive = rmin > 0
    max_negative = rmax < 0
    mid_range = np.logical_and(~min_positive, ~max_negative).astype(F32)
    min_positive = min_positive.astype(F32)
    max_negative = max_negative.astype(F32)

    levels = F32(2 ** n_bits - 1)
    with np.errstate(all="ignore"):
        scale = (rmax - rmin) / levels
        min_adj = scale * np.rint(rmin / scale)
        max_adj = rmax - rmin + min_adj
        min_adj = min_adj * mid_range + max_negative * rmin
        max_adj = max_adj * mid_range + min_positive * rmax
    return min_adj.astype(F32), max_adj.astype(F32)


def weights_uniform_params(num_bits: int, min_range, max_range):
    """(scales fp32[C], zero_points int32[C], qmin, qmax, adj_min fp32, adj_max fp32).

    weights_uniform_inferable_quantizer.py:123-124: the zero point is the
    TRUNCATION toward zero of min/scale, negated (torch ``.int()``).
    """
    a, b = fix_range_to_include_zero(min_range, max_range, num_bits)
    scales = ((b - a) / F32(2 ** num_bits - 1)).astype(F32)
    with np.errstate(all="ignore"):
        zps = (-np.trunc(a / scales)).astype(np.int32)
    return scales, zps, 0, 2 ** num_bits - 1, a, b


def activation_uniform_params(num_bits: int, min_range, max_range):
    """(scale python float, zero_point int, qmin, qmax, adj_min float, adj_max float).

    activation_uniform_inferable_quantizer.py:104-108: the adjusted range is read
    back as Python floats (exact float32 values), the scale is formed in double
    and the zero point is round-half-even of min/scale in double.
    """
    a, b = fix_range_to_include_zero(min_range, max_range, num_bits)
    a = float(a[0])
    b = float(b[0])
    scale = float((b - a) / ((2 ** num_bits) - 1))
    zp = int(-np.round(a / scale))
    return scale, zp, 0, 2 ** num_bits - 1, a, b


def validate_lut(num_bits, lut_values, threshold, signed, lut_values_bitwidth):
    """Constructor asserts of base_lut_symmetric_inferable_quantizer.py:53-86 (messages verbatim)."""
    assert isinstance(threshold, list), f'Threshold is expected to be a list, but is of type {type(threshold)}'
    assert isinstance(lut_values, list), f'lut_values is expected to be a list, but is of type {type(lut_values)}'
    lut = np.asarray(lut_values)
    assert len(np.unique(lut)) <= 2 ** num_bits, \
        f'Expected num of lut values to be less or equal than {2 ** num_bits} but got {len(lut)}'
    assert not np.any(lut - lut.astype(int)), 'Expected lut values to be integers'
    if signed:
        k = lut_values_bitwidth - 1
        assert np.all((-(2 ** k) <= lut) & (lut <= 2 ** k - 1)), 'Expected lut values in the quantization range'
    else:
        assert np.all(lut <= 2 ** lut_values_bitwidth), 'Expected lut values in the quantization range'
        assert np.all(lut >= 0), 'Expected unsigned lut values in unsigned activation quantization'
    assert num_bits <= lut_values_bitwidth, \
        f'Look-Up-Table bit configuration has {num_bits} bits. It must be less then {lut_values_bitwidth}'


# --------------------------------------------------------------------------
# half-precision storage (float16 / bfloat16 tensors; arithmetic stays float32)
# --------------------------------------------------------------------------

def narrow(a, dtype: str) -> np.ndarray:
    """Round float32 values to ``dtype`` (round-to-nearest-even) and widen back to float32."""
    a = np.asarray(a, dtype=F32)
    if dtype == "float32":
        return a
    if dtype == "float16":
        with np.errstate(all="ignore"):
            return a.astype(np.float16).astype(F32)
    if dtype == "bfloat16":
        u = a.view(np.uint32).astype(np.uint64)
        r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
        out = (r & 0xFFFFFFFF).astype(np.uint32).view(F32)
        return np.where(np.isnan(a), a, out).astype(F32)
    raise KeyError(dtype)


# --------------------------------------------------------------------------
# element arithmetic (__call__)
# --------------------------------------------------------------------------

def _channel_shape(ndim: int, axis: int):
    shape = [1] * ndim
    shape[axis] = -1
    return shape


def fake_quant_affine(x: np.ndarray, scale, zero_point, qmin: int, qmax: int,
                      axis=None, return_index: bool = False):
    """ATen fake_quantize_per_{tensor,channel}_affine on CPU, float32.

    Call sites: weights_symmetric_inferable_quantizer.py:139-151,
    weights_uniform_inferable_quantizer.py:153-165,
    activation_symmetric_inferable_quantizer.py:113-117,
    activation_uniform_inferable_quantizer.py:124-128.

    ``scale`` is a float32 vector (per-channel, ``axis`` given), a 1-element
    vector or a Python float (per-tensor; cast to float32 first, as ATen does).
    Returns y (and the integer clamp index q when ``return_index``).
    Defined for finite inputs with |x/scale| < 2**31 (see DESIGN.md for the
    saturation rule outside that domain).
    """
    x = np.asarray(x, dtype=F32)
    s = np.asarray(scale, dtype=F32).reshape(-1)
    z = np.asarray(zero_point).reshape(-1).astype(F32)
    if axis is not None:
        shape = _channel_shape(x.ndim, axis)
        s = s.reshape(shape)
        z = z.reshape(shape)
    else:
        assert s.size == 1 and z.size == 1
        s = s[0]
        z = z[0]
    with np.errstate(all="ignore"):
        inv = F32(1.0) / s                       # NOT x / s
        q = np.rint(x * inv) + z                 # ties to even, float add of the zero point
        q = np.minimum(np.maximum(q, F32(qmin)), F32(qmax))
        y = ((q - z) * s).astype(F32)
    if return_index:
        return y, q.astype(np.int64)
    return y


def lut_quantize(x: np.ndarray, lut_values, threshold, signed: bool,
                 lut_values_bitwidth: int = LUT_VALUES_BITWIDTH, eps: float = EPS,
                 per_channel: bool = False, channel_axis=None,
                 return_index: bool = False, chunk_elems: int = 1 << 20, step_dtype: str = "float32"):
    """pytorch/quantizer_utils.py:95-139 (lut_quantizer) + :142-170, float32.

    ``threshold`` is a float32 vector (weights, per-channel or 1 element) or a
    Python float (ActivationLutPOT, activation_lut_pot_inferable_quantizer.py:72):
    in that case ``threshold + eps`` is formed in double and then used as a
    float32 scalar.  The codebook scan is the literal first-minimum argmin over
    float32 |t - lut[j]| in list order (quantizer_utils.py:131-134).
    Processes the tensor in chunks so the N x L temporaries stay small.

    ``step_dtype`` ("float16"/"bfloat16", Python-float threshold only): the tensor is a half-precision
    activation, so ATen narrows the scalar divisor, the quotient and the scaled value to that type
    before the float32 codebook promotes the rest of the chain to float32.
    """
    x = np.asarray(x, dtype=F32)
    lut = np.asarray(lut_values, dtype=F32).reshape(-1)
    k = lut_values_bitwidth - int(signed)
    m = F32(2 ** k)
    if signed:
        cmin, cmax = F32(-2 ** (lut_values_bitwidth - 1)), F32(2 ** (lut_values_bitwidth - 1) - 1)
    else:
        cmin, cmax = F32(0), F32(2 ** lut_values_bitwidth - 1)

    if step_dtype != "float32":
        # torch.clip(half_tensor, min=python_float, max=python_float) converts the bounds to the tensor's type
        # (quantizer_utils.py:129): 2^k - 1 is not exact in bfloat16 beyond 8 bits, nor in float16 beyond 11, and
        # 65535 does not fit float16 at all -- torch raises then (pinned by tests/golden/cases_half_bounds.*)
        if step_dtype == "float16" and max(abs(float(cmin)), abs(float(cmax))) > 65504.0:
            raise RuntimeError("value cannot be converted to type c10::Half without overflow")
        cmin, cmax = (F32(narrow(np.asarray([b], dtype=F32), step_dtype)[0]) for b in (cmin, cmax))

    if isinstance(threshold, (float, int)):
        thr_mul = np.broadcast_to(F32(threshold), x.shape)
        div = F32(float(threshold) + eps)                                  # double add, then fp32
        if step_dtype != "float32":
            div = narrow(div, step_dtype).reshape(())[()]                  # ... then the tensor's type
        thr_div = np.broadcast_to(div, x.shape)
    else:
        thr = np.asarray(threshold, dtype=F32).reshape(-1)
        if per_channel:
            thr = thr.reshape(_channel_shape(x.ndim, channel_axis))
        thr_mul = np.broadcast_to(thr, x.shape)
        thr_div = np.broadcast_to((thr + F32(eps)).astype(F32), x.shape)   # fp32 add

    xf = x.reshape(-1)
    tm = thr_mul.reshape(-1)
    td = thr_div.reshape(-1)
    y = np.empty(xf.shape, dtype=F32)
    idx_out = np.empty(xf.shape, dtype=np.int64) if return_index else None
    with np.errstate(all="ignore"):
        for lo in range(0, xf.size, chunk_elems):
            hi = min(lo + chunk_elems, xf.size)
            if step_dtype == "float32":
                t = (xf[lo:hi] / td[lo:hi]) * m
            else:
                t = narrow(narrow(xf[lo:hi] / td[lo:hi], step_dtype) * m, step_dtype)
            t = np.where(np.isnan(t), t, np.minimum(np.maximum(t, cmin), cmax))  # torch.clip keeps NaN
            d = np.abs(t[:, None] - lut[None, :])
            # torch.argmin: first minimum, NaN counts as the minimum -> index 0 for an all-NaN row
            idx = np.where(np.isnan(t), 0, np.argmin(d, axis=1))
            y[lo:hi] = (lut[idx] / m) * tm[lo:hi]
            if return_index:
                idx_out[lo:hi] = idx
    y = y.reshape(x.shape)
    if return_index:
        return y, idx_out.reshape(x.shape)
    return y


# --------------------------------------------------------------------------
# float64 tensors (the reference passes them to ATen unchanged; same call sites)
# --------------------------------------------------------------------------

def fake_quant_affine_f64(x: np.ndarray, scale, zero_point, qmin: int, qmax: int, axis=None):
    """ATen's CPU fake-quant kernels instantiated for double (same call sites as fake_quant_affine).

    The clamp index uses a DOUBLE product with the float32 reciprocal widened:
        q = clamp(nearbyint(x * (double)(1.0f / s)) + z, qmin, qmax)
    The dequantized value differs by overload (pinned by tests/golden/cases_f64.*):
        per tensor (float or 1-element-tensor qparams):  y = (double)((float)(q - z) * s)
        per channel:                                     y = (double)(q - z) * (double)s
    """
    x = np.asarray(x, dtype=np.float64)
    s = np.asarray(scale, dtype=F32).reshape(-1)
    z = np.asarray(zero_point).reshape(-1).astype(np.float64)
    if axis is not None:
        shape = _channel_shape(x.ndim, axis)
        s = s.reshape(shape)
        z = z.reshape(shape)
    else:
        assert s.size == 1 and z.size == 1
        s = s[0]
        z = z[0]
    with np.errstate(all="ignore"):
        inv = (F32(1.0) / s).astype(np.float64)
        q = np.rint(x * inv) + z
        q = np.minimum(np.maximum(q, float(qmin)), float(qmax))
        if axis is not None:
            return (q - z) * s.astype(np.float64)
        return ((q - z).astype(F32) * s).astype(np.float64)


def lut_quantize_f64(x: np.ndarray, lut_values, threshold, signed: bool,
                     lut_values_bitwidth: int = LUT_VALUES_BITWIDTH, eps: float = EPS,
                     per_channel: bool = False, channel_axis=None, chunk_elems: int = 1 << 19):
    """pytorch/quantizer_utils.py:95-139 on a float64 tensor: type promotion makes the quotient, the clip and the
    distances double, while (lut[idx] / 2^k) * threshold stays float32 -- the OUTPUT is float32.
    ``threshold`` float32 vector (weights: divisor = fl32(thr + fl32(eps)) widened) or Python float (activation:
    divisor = thr + eps, a double)."""
    x = np.asarray(x, dtype=np.float64)
    lut = np.asarray(lut_values, dtype=F32).reshape(-1)
    k = lut_values_bitwidth - int(signed)
    m = float(2 ** k)
    if signed:
        cmin, cmax = float(-2 ** (lut_values_bitwidth - 1)), float(2 ** (lut_values_bitwidth - 1) - 1)
    else:
        cmin, cmax = 0.0, float(2 ** lut_values_bitwidth - 1)
    if isinstance(threshold, (float, int)):
        thr_mul = np.broadcast_to(F32(threshold), x.shape)
        thr_div = np.broadcast_to(np.float64(float(threshold) + eps), x.shape)
    else:
        thr = np.asarray(threshold, dtype=F32).reshape(-1)
        if per_channel:
            thr = thr.reshape(_channel_shape(x.ndim, channel_axis))
        thr_mul = np.broadcast_to(thr, x.shape)
        thr_div = np.broadcast_to((thr + F32(eps)).astype(F32).astype(np.float64), x.shape)
    xf, tm, td = x.reshape(-1), thr_mul.reshape(-1), thr_div.reshape(-1)
    y = np.empty(xf.shape, dtype=F32)
    lut64 = lut.astype(np.float64)
    with np.errstate(all="ignore"):
        for lo in range(0, xf.size, chunk_elems):
            hi = min(lo + chunk_elems, xf.size)
            t = (xf[lo:hi] / td[lo:hi]) * m
            t = np.where(np.isnan(t), t, np.minimum(np.maximum(t, cmin), cmax))
            d = np.abs(t[:, None] - lut64[None, :])
            idx = np.where(np.isnan(t), 0, np.argmin(d, axis=1))
            y[lo:hi] = (lut[idx] / F32(m)) * tm[lo:hi]
    return y.reshape(x.shape)


# --------------------------------------------------------------------------
# export-time arithmetic (`_use_custom_impl and torch.jit.is_tracing()`)
# --------------------------------------------------------------------------

def export_grid(x: np.ndarray, lo, hi, step, axis=None, shifted: bool = False) -> np.ndarray:
    """clip -> TRUE division -> round half even -> scale back, float32 throughout.

    weights_symmetric_inferable_quantizer.py:67-68 and weights_uniform_inferable_quantizer.py:75-77
    (``torch.where`` clip, ``round(c / step) * step``); activation_symmetric_inferable_quantizer.py:53
    (``torch.clip``, same formula); activation_uniform_inferable_quantizer.py:60-64
    (``step * round((c - lo) / step) + lo``).  NaN passes through; a value equal to a bound keeps its own
    sign of zero.  ``lo`` / ``hi`` / ``step``: float32 vectors along ``axis`` or scalars (rounded to float32).
    """
    x = np.asarray(x, dtype=F32)
    lo, hi, step = (np.asarray(v, dtype=np.float64).astype(F32) for v in (lo, hi, step))
    if axis is not None:
        shape = _channel_shape(x.ndim, axis)
        lo, hi, step = lo.reshape(shape), hi.reshape(shape), step.reshape(shape)
    else:
        lo, hi, step = lo.reshape(-1)[0], hi.reshape(-1)[0], step.reshape(-1)[0]
    with np.errstate(all="ignore"):
        c = np.where(x < lo, lo, x).astype(F32)
        c = np.where(x > hi, hi, c).astype(F32)
        if shifted:
            return ((step * np.rint(((c - lo).astype(F32) / step).astype(F32))).astype(F32) + lo).astype(F32)
        return (np.rint((c / step).astype(F32)) * step).astype(F32)


def export_weights_symmetric(x, num_bits: int, threshold, axis=None):
    """quantize_sym_weights_torch (weights_symmetric_inferable_quantizer.py:32-70): float32 parameter math."""
    thr = np.asarray(threshold, dtype=np.float64).astype(F32)
    step = (thr / F32(2 ** (num_bits - 1))).astype(F32)
    return export_grid(x, -thr, (thr - step).astype(F32), step, axis)


def export_weights_uniform(x, num_bits: int, adj_min, adj_max, axis=None):
    """quantize_uniform_weights_torch (weights_uniform_inferable_quantizer.py:34-78) on the quantizer's
    ALREADY adjusted ranges (``:143-148`` passes adjusted_{min,max}_range_np); the range fix runs again."""
    a, b = fix_range_to_include_zero(adj_min, adj_max, num_bits)
    step = ((b - a) / F32(2 ** num_bits - 1)).astype(F32)
    return export_grid(x, a, b, step, axis)


def export_activation_symmetric(x, num_bits: int, threshold: float, signed: bool):
    """quantize_sym_activations_torch (activation_symmetric_inferable_quantizer.py:29-54): double parameters."""
    threshold = float(threshold)
    if signed:
        step = threshold / (2 ** (num_bits - 1))
        lo, hi = -threshold, threshold - step
    else:
        step = threshold / (2 ** num_bits)
        lo, hi = 0.0, threshold - step
    return export_grid(x, lo, hi, step)


def adjust_range_to_include_zero_f64(range_min: float, range_max: float, n_bits: int):
    """common/quant_utils.py:20-50 on Python floats (double arithmetic, final clamp to lo <= 0 <= hi)."""
    rmin, rmax = np.float64(range_min), np.float64(range_max)
    scale = (rmax - rmin) / (2 ** n_bits - 1)
    a = scale * np.round(rmin / scale)
    b = rmax - rmin + a
    pos, neg = rmin > 0, rmax < 0
    mid = (not pos) and (not neg)
    a = a * mid + neg * rmin
    b = b * mid + pos * rmax
    return float(np.minimum(a, 0)), float(np.maximum(b, 0))


def export_activation_uniform(x, num_bits: int, min_range: float, max_range: float):
    """quantize_uniform_activations_torch (activation_uniform_inferable_quantizer.py:32-65) on the
    quantizer's adjusted Python-float range (``:121``)."""
    a, b = adjust_range_to_include_zero_f64(min_range, max_range, num_bits)
    step = (b - a) / (2 ** num_bits - 1)
    return export_grid(x, a, b, step, shifted=True)


# --------------------------------------------------------------------------
# integer consumer of the codes (extension: no counterpart in the reference)
# --------------------------------------------------------------------------

def qlinear_i8(a_codes: np.ndarray, a_zero_point: int, a_scale: float, w_codes: np.ndarray, w_scales,
               bias=None) -> np.ndarray:
    """y[m][n] = float32(sum_k (a[m][k] - za) * w[n][k]) * (float32(sa) * sw[n]) (+ bias[n]); exact integer sum,
    one rounding per float32 operation -- the contract of include/mctq_hip.h: mctq_qlinear_i8.

    It evaluates what PytorchQuantizationWrapper.forward (quantize_wrapper.py:231-257) computes for a wrapped
    torch.nn.Linear fed by an activation holder, F.linear(fake_quant(x), fake_quant(W), bias), on the clamp
    indices instead of the dequantized float32 values (the two differ by float32 rounding of the long sum)."""
    a64 = np.asarray(a_codes).astype(np.int64) - int(a_zero_point)
    w64 = np.asarray(w_codes).astype(np.int64)
    if a64.shape[0] * w64.shape[0] * a64.shape[1] <= 1 << 24:
        acc = a64 @ w64.T
    else:
        # large cases: the same integer sum through float64 BLAS -- exact, every partial sum is an integer below
        # 2^53 (|a - za| <= 383, |w| <= 128, K <= 32768), whatever the summation order; tests/test_consumers.py checks
        # the two branches against each other
        acc = np.rint(a64.astype(np.float64) @ w64.astype(np.float64).T).astype(np.int64)
    assert np.all(np.abs(acc) < 2 ** 31)
    sc = (F32(a_scale) * np.asarray(w_scales, dtype=F32).reshape(-1)).astype(F32)
    y = (acc.astype(np.int32).astype(F32) * sc[None, :]).astype(F32)
    if bias is not None:
        y = (y + np.asarray(bias, dtype=F32)[None, :]).astype(F32)
    return y


def pack4(q: np.ndarray) -> np.ndarray:
    """4-bit code packing of include/mctq_hip.h (MCTQ_CODE_I4 / _U4): integer codes in storage order, element 2j in
    the low nibble of byte j, element 2j + 1 in the high nibble; signed codes as two's-complement nibbles."""
    flat = np.asarray(q).reshape(-1).astype(np.int64) & 0xF
    return (flat[0::2] | (flat[1::2] << 4)).astype(np.uint8)
