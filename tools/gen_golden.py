#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the REFERENCE implementation.

Runs only in the build container: imports sony/mct_quantizers from /root/reference
(torch CPU) and records, for every quantizer class on the hot path,

  * ``cases.json`` + ``cases.npz`` : (constructor kwargs, input) -> output triples on
    adversarial inputs (round-half ties, +-threshold, beyond the clip range, +-0,
    denormals, large finite values) for per-tensor / axis 0 / middle axis / last axis,
    contiguous and channels_last layouts, bits {2,3,4,8};
  * ``ctor.json``                  : constructor-derived attributes (scales, zero points,
    adjusted ranges, clamp domains), incl. the trunc-vs-round zero-point cases;
  * ``errors.json``                : the verbatim assertion messages of illegal constructions;
  * ``full_sha.json``              : SHA-256 of the reference outputs at the full BASELINE
    sizes for the portable synthetic inputs of mct_quantizers_amd/workloads.py.

The fixtures are data only (inputs and expected outputs).  The reference source never
enters this repository.  Usage:  python tools/gen_golden.py [--skip-full]
"""
import argparse
import hashlib
import json
import os
import sys
import warnings

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, "/root/reference")
sys.path.insert(1, REPO)

import torch  # noqa: E402
import mct_quantizers as ref  # noqa: E402  (the reference)
from mct_quantizers.pytorch import quantizers as refq  # noqa: E402

# workloads.py only needs numpy; load it without importing the package under test
import importlib.util  # noqa: E402
_spec = importlib.util.spec_from_file_location("workloads", os.path.join(REPO, "mct_quantizers_amd", "workloads.py"))
workloads = importlib.util.module_from_spec(_spec)
sys.modules["workloads"] = workloads
_spec.loader.exec_module(workloads)

OUT = os.path.join(REPO, "tests", "golden")
rng = np.random.default_rng(20251003)


def adversarial(shape, scales, axis, qmin, qmax, zps=None):
    """Input whose entries hit every interesting point of the quantization grid of their channel."""
    n = int(np.prod(shape))
    x = rng.standard_normal(n).astype(np.float32).reshape(shape)
    s = np.asarray(scales, dtype=np.float32).reshape(-1)
    z = np.zeros_like(s) if zps is None else np.asarray(zps, dtype=np.float32).reshape(-1)
    if axis is None:
        sb = np.broadcast_to(s[0], shape)
        zb = np.broadcast_to(z[0], shape)
    else:
        bs = [1] * len(shape)
        bs[axis] = -1
        sb = np.broadcast_to(s.reshape(bs), shape)
        zb = np.broadcast_to(z.reshape(bs), shape)
    x = x * sb * np.float32(0.35 * (qmax - qmin))
    flat = x.reshape(-1)
    sf = sb.reshape(-1)
    zf = zb.reshape(-1)
    kinds = rng.integers(0, 12, size=n)
    k = rng.integers(qmin - 3, qmax + 4, size=n).astype(np.float32) - zf
    flat = np.where(kinds == 0, (k + np.float32(0.5)) * sf, flat)                 # exact ties
    flat = np.where(kinds == 1, np.nextafter((k + np.float32(0.5)) * sf, np.float32(np.inf)), flat)
    flat = np.where(kinds == 2, np.nextafter((k + np.float32(0.5)) * sf, np.float32(-np.inf)), flat)
    flat = np.where(kinds == 3, (np.float32(qmax) - zf) * sf, flat)               # upper edge
    flat = np.where(kinds == 4, (np.float32(qmin) - zf) * sf, flat)               # lower edge
    flat = np.where(kinds == 5, k * sf, flat)                                     # on-grid
    flat = flat.astype(np.float32)
    special = np.array([0.0, -0.0, 1e-45, -1e-45, 1e-39, -1e-39, 1.17549435e-38, 3.0e5, -3.0e5], dtype=np.float32)
    pos = rng.choice(n, size=min(n, special.size * 2), replace=False)
    flat[pos] = np.resize(special, pos.size)
    # large finite values, kept inside |x/s| < 2**31
    pos = rng.choice(n, size=max(1, n // 50), replace=False)
    flat[pos] = (sf[pos] * np.float32(2.0 ** 30) * rng.choice([-1.0, 1.0], size=pos.size)).astype(np.float32)
    return flat.reshape(shape).astype(np.float32)


def to_jsonable(v):
    if isinstance(v, np.ndarray):
        return v.tolist()
    if isinstance(v, (np.floating,)):
        return float(v)
    if isinstance(v, (np.integer,)):
        return int(v)
    if isinstance(v, torch.Tensor):
        return v.detach().cpu().numpy().tolist()
    return v


cases = []
arrays = {}


def add_case(cls_name, kwargs, x, memory_format=None, note=""):
    cls = getattr(refq, cls_name)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        q = cls(**kwargs)
    xt = torch.from_numpy(np.ascontiguousarray(x))
    if memory_format == "channels_last":
        xt = xt.contiguous(memory_format=torch.channels_last)
    elif memory_format == "transposed":
        xt = xt.transpose(0, -1).contiguous().transpose(0, -1)
    y = q(xt)
    assert y.shape == xt.shape
    cid = f"c{len(cases):03d}"
    arrays[cid + "_x"] = x
    arrays[cid + "_y"] = y.detach().numpy().copy()
    cases.append(dict(id=cid, cls=cls_name, kwargs=kwargs, shape=list(x.shape),
                      memory_format=memory_format, out_strides=list(y.stride()), note=note))


def gen_affine_cases():
    for bits in (2, 3, 4, 8):
        qmin, qmax = -2 ** (bits - 1), 2 ** (bits - 1) - 1
        # weights symmetric per tensor (tensor-qparams overload)
        thr = [float(rng.uniform(0.3, 5.0))]
        sc = np.asarray(thr) / 2 ** (bits - 1)
        add_case("WeightsSymmetricInferableQuantizer", dict(num_bits=bits, threshold=thr, per_channel=False),
                 adversarial((7, 33, 5), sc, None, qmin, qmax))
        # per channel: axis 0 / middle / last, odd sizes
        for shape, axis in (((6, 37), 0), ((5, 7, 11), 1), ((1, 10, 10, 3), 3), ((4, 8, 16), 2), ((3, 5, 4, 4), 1)):
            C = shape[axis]
            thr = [float(v) for v in rng.uniform(0.05, 7.0, size=C)]
            sc = np.asarray(thr) / 2 ** (bits - 1)
            add_case("WeightsSymmetricInferableQuantizer",
                     dict(num_bits=bits, threshold=thr, per_channel=True, channel_axis=axis),
                     adversarial(shape, sc, axis, qmin, qmax))
        # POT per channel
        thr = [float(2.0 ** e) for e in rng.integers(-4, 4, size=6)]
        sc = np.asarray(thr) / 2 ** (bits - 1)
        add_case("WeightsPOTInferableQuantizer", dict(num_bits=bits, threshold=thr, per_channel=True, channel_axis=0),
                 adversarial((6, 40), sc, 0, qmin, qmax))
        add_case("WeightsPOTInferableQuantizer", dict(num_bits=bits, threshold=[2.0], per_channel=False),
                 adversarial((9, 9), np.asarray([2.0]) / 2 ** (bits - 1), None, qmin, qmax))
        # weights uniform per tensor / per channel
        uq = refq.WeightsUniformInferableQuantizer
        for shape, axis in (((6, 37), 0), ((5, 7, 11), 1), ((2, 9, 9, 5), 3)):
            C = shape[axis]
            lo = [float(v) for v in rng.uniform(-4.0, 0.5, size=C)]
            hi = [float(a + d) for a, d in zip(lo, rng.uniform(0.2, 6.0, size=C))]
            kw = dict(num_bits=bits, min_range=lo, max_range=hi, per_channel=True, channel_axis=axis)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                q = uq(**kw)
            add_case("WeightsUniformInferableQuantizer", kw,
                     adversarial(shape, q.scales.numpy().reshape(-1), axis, 0, 2 ** bits - 1,
                                 q.zero_points.numpy().reshape(-1)))
        kw = dict(num_bits=bits, min_range=[-1.3], max_range=[2.9], per_channel=False)
        q = uq(**kw)
        add_case("WeightsUniformInferableQuantizer", kw,
                 adversarial((11, 13), q.scales.numpy(), None, 0, 2 ** bits - 1, q.zero_points.numpy()))
        # activations
        for signed in (True, False):
            thr = [float(rng.uniform(0.5, 6.0))]
            sc = np.asarray(thr) / (2 ** (bits - 1) if signed else 2 ** bits)
            dom = (qmin, qmax) if signed else (0, 2 ** bits - 1)
            add_case("ActivationSymmetricInferableQuantizer", dict(num_bits=bits, threshold=thr, signed=signed),
                     adversarial((2, 3, 17, 9), sc, None, *dom))
            add_case("ActivationPOTInferableQuantizer", dict(num_bits=bits, threshold=[4.0], signed=signed),
                     adversarial((3, 50), np.asarray([4.0]) / (2 ** (bits - 1) if signed else 2 ** bits), None, *dom))
        for lo, hi in ((-2.5, 3.1), (3.0, 10.0), (-7.0, -1.0), (-0.37, 0.91)):
            kw = dict(num_bits=bits, min_range=[lo], max_range=[hi])
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                q = refq.ActivationUniformInferableQuantizer(**kw)
            add_case("ActivationUniformInferableQuantizer", kw,
                     adversarial((2, 3, 12, 12), [q.scale], None, 0, 2 ** bits - 1, [q.zero_point]))
    # memory formats (ATen preserves strides)
    thr = [float(v) for v in rng.uniform(0.1, 3.0, size=5)]
    sc = np.asarray(thr) / 128
    add_case("WeightsSymmetricInferableQuantizer", dict(num_bits=8, threshold=thr, per_channel=True, channel_axis=1),
             adversarial((2, 5, 6, 6), sc, 1, -128, 127), memory_format="channels_last")
    thr = [float(v) for v in rng.uniform(0.1, 3.0, size=6)]
    sc = np.asarray(thr) / 128
    add_case("WeightsSymmetricInferableQuantizer", dict(num_bits=8, threshold=thr, per_channel=True, channel_axis=0),
             adversarial((6, 20), sc, 0, -128, 127), memory_format="transposed")
    add_case("ActivationSymmetricInferableQuantizer", dict(num_bits=8, threshold=[3.0], signed=True),
             adversarial((2, 4, 5, 5), [3.0 / 128], None, -128, 127), memory_format="channels_last")
    # empty and 0-dim-like
    add_case("ActivationSymmetricInferableQuantizer", dict(num_bits=8, threshold=[3.0], signed=True),
             np.zeros((0, 4), dtype=np.float32), note="empty")
    add_case("WeightsSymmetricInferableQuantizer", dict(num_bits=8, threshold=[1.0, 2.0], per_channel=True, channel_axis=1),
             np.zeros((0, 2), dtype=np.float32), note="empty")
    add_case("ActivationSymmetricInferableQuantizer", dict(num_bits=8, threshold=[3.0], signed=True),
             np.asarray([1.2345], dtype=np.float32), note="single element")


def lut_input(shape, thr, axis, span=1.6):
    n = int(np.prod(shape))
    x = rng.standard_normal(n).astype(np.float32).reshape(shape) * np.float32(0.5)
    t = np.asarray(thr, dtype=np.float32).reshape(-1)
    if axis is None or t.size == 1:
        tb = np.broadcast_to(t[0], shape)
    else:
        bs = [1] * len(shape)
        bs[axis] = -1
        tb = np.broadcast_to(t.reshape(bs), shape)
    x = (x * tb * np.float32(span)).astype(np.float32)
    flat = x.reshape(-1)
    tf = tb.reshape(-1)
    # exact midpoints between integer codes, and +-threshold
    pos = rng.choice(n, size=max(1, n // 6), replace=False)
    codes = rng.integers(-130, 130, size=pos.size).astype(np.float32) + np.float32(0.5)
    flat[pos] = (codes / np.float32(128.0) * tf[pos]).astype(np.float32)
    special = np.array([0.0, -0.0, 1e-45, 1e-39, -1e-39, 1e-9, -1e-9, 3.0e5, -3.0e5, 1e30, -1e30], dtype=np.float32)
    pos = rng.choice(n, size=min(n, special.size), replace=False)
    flat[pos] = special[:pos.size]
    return flat.reshape(shape).astype(np.float32)


def gen_lut_cases():
    lut8 = [22.0, -53.0, 62.0, 0.0, -66.0, -21.0, 44.0, -40.0]            # unsorted (compat tests)
    lut16 = list(workloads.CFG4_LUT)
    lut4 = [-25.0, 25.0, -100.0, 100.0]
    for lut, bits in ((lut8, 3), (lut16, 4), (lut4, 2), ([-5.0, 5.0], 1), ([7.0], 2), ([3.0, 3.0, -8.0, 3.0], 2)):
        thr = [float(rng.uniform(0.5, 3.0))]
        add_case("WeightsLUTSymmetricInferableQuantizer",
                 dict(num_bits=bits, lut_values=lut, threshold=thr, per_channel=False),
                 lut_input((9, 31), thr, None))
        for shape, axis in (((6, 37), 0), ((5, 7, 11), 1), ((1, 10, 10, 3), 3)):
            C = shape[axis]
            thr = [float(v) for v in rng.uniform(0.05, 4.0, size=C)]
            add_case("WeightsLUTSymmetricInferableQuantizer",
                     dict(num_bits=bits, lut_values=lut, threshold=thr, per_channel=True, channel_axis=axis,
                          input_rank=len(shape)),
                     lut_input(shape, thr, axis))
        thr = [float(2.0 ** e) for e in rng.integers(-3, 3, size=6)]
        add_case("WeightsLUTPOTInferableQuantizer",
                 dict(num_bits=bits, lut_values=lut, threshold=thr, per_channel=True, channel_axis=0, input_rank=2),
                 lut_input((6, 29), thr, 0))
        add_case("ActivationLutPOTInferableQuantizer",
                 dict(num_bits=bits, lut_values=lut, threshold=[2.0], signed=True),
                 lut_input((2, 3, 8, 8), [2.0], None))
    # unsigned activation LUT, non-default bitwidth / eps, tiny POT threshold (eps matters)
    add_case("ActivationLutPOTInferableQuantizer",
             dict(num_bits=3, lut_values=[0.0, 13.0, 50.0, 90.0, 128.0, 200.0, 255.0, 256.0], threshold=[4.0], signed=False),
             np.abs(lut_input((3, 64), [4.0], None)) * np.float32(1.3))
    add_case("ActivationLutPOTInferableQuantizer",
             dict(num_bits=2, lut_values=[-8.0, -3.0, 2.0, 7.0], threshold=[2.0 ** -20], signed=True,
                  lut_values_bitwidth=4, eps=1e-8),
             lut_input((4, 40), [2.0 ** -20], None))
    add_case("WeightsLUTSymmetricInferableQuantizer",
             dict(num_bits=4, lut_values=[float(v) for v in range(-512, 512, 64)], threshold=[0.7, 1.9, 0.33],
                  per_channel=True, channel_axis=1, input_rank=3, lut_values_bitwidth=10, eps=1e-3),
             lut_input((4, 3, 21), [0.7, 1.9, 0.33], 1))
    # channels_last per-channel LUT: the reference's output is contiguous in whatever layout
    # broadcasting gives; record strides
    thr = [float(v) for v in rng.uniform(0.1, 3.0, size=5)]
    add_case("WeightsLUTSymmetricInferableQuantizer",
             dict(num_bits=4, lut_values=lut16, threshold=thr, per_channel=True, channel_axis=1, input_rank=4),
             lut_input((2, 5, 6, 6), thr, 1), memory_format="channels_last")
    # 8-bit LUT with 256 entries (largest legal codebook at the default bitwidth)
    lut256 = [float(v) for v in rng.permutation(np.arange(-128, 128))]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        add_case("WeightsLUTSymmetricInferableQuantizer",
                 dict(num_bits=8, lut_values=lut256, threshold=[1.5], per_channel=False),
                 lut_input((8, 16), [1.5], None))
    add_case("WeightsLUTSymmetricInferableQuantizer",
             dict(num_bits=4, lut_values=lut16, threshold=[1.5], per_channel=False),
             np.zeros((0, 3), dtype=np.float32), note="empty")


def gen_ctor():
    out = []

    def rec(cls_name, kwargs, attrs):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            q = getattr(refq, cls_name)(**kwargs)
        d = {}
        for a in attrs:
            v = getattr(q, a)
            if isinstance(v, torch.Tensor):
                d[a] = dict(dtype=str(v.dtype), values=[float(t) for t in v.reshape(-1).double()])
            elif isinstance(v, np.ndarray):
                d[a] = dict(dtype=str(v.dtype), values=[float(t) for t in v.reshape(-1).astype(np.float64)])
            else:
                d[a] = to_jsonable(v)
        out.append(dict(cls=cls_name, kwargs=kwargs, attrs=d))

    for bits in (2, 3, 4, 5, 7, 8):
        thr = [float(v) for v in rng.uniform(0.01, 9.0, size=7)]
        rec("WeightsSymmetricInferableQuantizer", dict(num_bits=bits, threshold=thr, per_channel=True, channel_axis=0),
            ["scales", "zero_points", "min_quantized_domain", "max_quantized_domain", "num_bits", "signed"])
        rec("WeightsPOTInferableQuantizer", dict(num_bits=bits, threshold=[0.5, 4.0], per_channel=True, channel_axis=1),
            ["scales", "zero_points", "min_quantized_domain", "max_quantized_domain"])
        for signed in (True, False):
            rec("ActivationSymmetricInferableQuantizer", dict(num_bits=bits, threshold=[thr[0]], signed=signed),
                ["scales", "zero_points", "min_quantized_domain", "max_quantized_domain", "threshold_np"])
        lo = [float(v) for v in rng.uniform(-5.0, 1.0, size=40)]
        hi = [float(a + d) for a, d in zip(lo, rng.uniform(0.1, 7.0, size=40))]
        rec("WeightsUniformInferableQuantizer",
            dict(num_bits=bits, min_range=lo, max_range=hi, per_channel=True, channel_axis=0),
            ["scales", "zero_points", "adjusted_min_range_np", "adjusted_max_range_np",
             "min_quantized_domain", "max_quantized_domain"])
        for a, b in ((-2.5, 3.1), (3.0, 10.0), (-7.0, -1.0), (-4.0, 4.0), (-3.0, 3.0), (-0.1234, 7.77)):
            rec("ActivationUniformInferableQuantizer", dict(num_bits=bits, min_range=[a], max_range=[b]),
                ["scale", "zero_point", "min_range", "max_range", "min_quantized_domain", "max_quantized_domain"])
    return out


def gen_errors():
    out = []

    def rec(cls_name, kwargs_repr, fn):
        try:
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                fn()
        except Exception as e:  # noqa: BLE001
            out.append(dict(cls=cls_name, kwargs=kwargs_repr, exc=type(e).__name__, msg=str(e)))
        else:
            raise RuntimeError(f"expected an error from {cls_name} {kwargs_repr}")

    def both(cls_name, **kw):
        rec(cls_name, kw, lambda: getattr(refq, cls_name)(**kw))

    both("WeightsSymmetricInferableQuantizer", num_bits=8, threshold=[3.0, 2.0], per_channel=False)
    both("WeightsSymmetricInferableQuantizer", num_bits=8, threshold=[3.0, 2.0], per_channel=True)
    both("WeightsSymmetricInferableQuantizer", num_bits=8, threshold=[], per_channel=True, channel_axis=0)
    both("WeightsPOTInferableQuantizer", num_bits=8, threshold=[3.0], per_channel=False)
    both("WeightsPOTInferableQuantizer", num_bits=8, threshold=[2.0, 3.0], per_channel=True, channel_axis=0)
    both("WeightsUniformInferableQuantizer", num_bits=8, min_range=[3.0, 2.0], max_range=[4.0, 3.0], per_channel=False)
    both("WeightsUniformInferableQuantizer", num_bits=8, min_range=[3.0], max_range=[1.0], per_channel=False)
    both("WeightsUniformInferableQuantizer", num_bits=8, min_range=[0.0], max_range=[1.0], per_channel=True)
    both("ActivationSymmetricInferableQuantizer", num_bits=8, threshold=[4.0, 2.0], signed=True)
    both("ActivationPOTInferableQuantizer", num_bits=8, threshold=[3.0], signed=True)
    both("ActivationUniformInferableQuantizer", num_bits=8, min_range=[0.0, 1.0], max_range=[2.0, 3.0])
    both("ActivationUniformInferableQuantizer", num_bits=8, min_range=[4.0], max_range=[2.0])
    lut = [-25.0, 25.0]
    both("WeightsLUTSymmetricInferableQuantizer", num_bits=1, lut_values=[-25.0, 25.0, 3.0], threshold=[2.0], per_channel=False)
    both("WeightsLUTSymmetricInferableQuantizer", num_bits=3, lut_values=[-25.5, 25.0], threshold=[2.0], per_channel=False)
    both("WeightsLUTSymmetricInferableQuantizer", num_bits=3, lut_values=[-250.0, 25.0], threshold=[2.0], per_channel=False)
    both("WeightsLUTSymmetricInferableQuantizer", num_bits=10, lut_values=lut, threshold=[2.0], per_channel=False)
    both("WeightsLUTSymmetricInferableQuantizer", num_bits=3, lut_values=lut, threshold=[2.0, 3.0], per_channel=False)
    both("WeightsLUTSymmetricInferableQuantizer", num_bits=3, lut_values=lut, threshold=[2.0], per_channel=True)
    both("WeightsLUTSymmetricInferableQuantizer", num_bits=3, lut_values=lut, threshold=[2.0], per_channel=True, channel_axis=0)
    both("WeightsLUTPOTInferableQuantizer", num_bits=3, lut_values=lut, threshold=[3.0], per_channel=False)
    both("ActivationLutPOTInferableQuantizer", num_bits=3, lut_values=lut, threshold=[3.0], signed=True)
    both("ActivationLutPOTInferableQuantizer", num_bits=3, lut_values=lut, threshold=[2.0, 4.0], signed=True)
    both("ActivationLutPOTInferableQuantizer", num_bits=3, lut_values=[-25.0, 25.0], threshold=[2.0], signed=False)
    both("ActivationLutPOTInferableQuantizer", num_bits=3, lut_values=[25.0, 300.0], threshold=[2.0], signed=False)
    # non-list arguments
    rec("WeightsSymmetricInferableQuantizer", "threshold=np.asarray([2.0])",
        lambda: refq.WeightsSymmetricInferableQuantizer(num_bits=8, threshold=np.asarray([2.0]), per_channel=False))
    rec("ActivationUniformInferableQuantizer", "min_range=np.asarray([0.0]), max_range=[1.0]",
        lambda: refq.ActivationUniformInferableQuantizer(num_bits=8, min_range=np.asarray([0.0]), max_range=[1.0]))
    rec("ActivationUniformInferableQuantizer", "min_range=[0.0], max_range=np.asarray([1.0])",
        lambda: refq.ActivationUniformInferableQuantizer(num_bits=8, min_range=[0.0], max_range=np.asarray([1.0])))
    rec("WeightsLUTSymmetricInferableQuantizer", "threshold=np.asarray([2.0])",
        lambda: refq.WeightsLUTSymmetricInferableQuantizer(num_bits=3, lut_values=lut, threshold=np.asarray([2.0]),
                                                           per_channel=False))
    rec("WeightsLUTSymmetricInferableQuantizer", "lut_values=np.asarray([-25, 25])",
        lambda: refq.WeightsLUTSymmetricInferableQuantizer(num_bits=3, lut_values=np.asarray(lut), threshold=[2.0],
                                                           per_channel=False))
    return out


def gen_full_sha():
    out = {}
    torch.set_num_threads(os.cpu_count() or 1)
    for cfg in ("cfg1", "cfg2", "cfg3", "cfg4", "cfg5"):
        x = workloads.make_input(cfg, batch=8)
        wl = workloads.make_workload(cfg, x)
        q = getattr(refq, wl.quantizer)(**wl.kwargs)
        y = q(torch.from_numpy(x)).numpy()
        out[cfg] = dict(shape=list(x.shape), quantizer=wl.quantizer,
                        x_sha256=hashlib.sha256(x.tobytes()).hexdigest(),
                        y_sha256=hashlib.sha256(np.ascontiguousarray(y).tobytes()).hexdigest(),
                        y_sum_f64=float(np.sum(y, dtype=np.float64)),
                        n_unique=int(min(np.unique(y[:64]).size, 1 << 20)))
        print(cfg, out[cfg], flush=True)
        del x, y
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--skip-full", action="store_true")
    ap.add_argument("--pickles-only", action="store_true")
    ap.add_argument("--half-only", action="store_true")
    ap.add_argument("--half-bounds-only", action="store_true")
    args = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    gen_affine_cases()
    gen_lut_cases()
    np.savez_compressed(os.path.join(OUT, "cases.npz"), **arrays)
    meta = dict(reference="sony/mct_quantizers v%s" % ref.__version__, torch=torch.__version__,
                numpy=np.__version__, generator="tools/gen_golden.py")
    with open(os.path.join(OUT, "cases.json"), "w") as f:
        json.dump(dict(meta=meta, cases=cases), f, indent=1)
    with open(os.path.join(OUT, "ctor.json"), "w") as f:
        json.dump(dict(meta=meta, ctor=gen_ctor()), f, indent=1)
    with open(os.path.join(OUT, "errors.json"), "w") as f:
        json.dump(dict(meta=meta, errors=gen_errors()), f, indent=1)
    if not args.skip_full:
        with open(os.path.join(OUT, "full_sha.json"), "w") as f:
            json.dump(dict(meta=meta, configs=gen_full_sha()), f, indent=1)
    print(f"{len(cases)} cases, {sum(a.nbytes for a in arrays.values())} array bytes")




def gen_half_cases():
    """float16 / bfloat16 inputs: ATen keeps the input type for the affine quantizers, the LUT chain promotes
    to float32.  Inputs/outputs are stored widened to float32 (exact) with their dtype names."""
    hc, ha = [], {}

    def add(cls_name, kwargs, x32, dt_name):
        dt = getattr(torch, dt_name)
        xt = torch.from_numpy(np.ascontiguousarray(x32)).to(dt)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            q = getattr(refq, cls_name)(**kwargs)
        y = q(xt.clone())
        cid = f"h{len(hc):03d}"
        ha[cid + "_x"] = xt.float().numpy()
        ha[cid + "_y"] = y.detach().float().numpy()
        hc.append(dict(id=cid, cls=cls_name, kwargs=kwargs, shape=list(x32.shape), in_dtype=dt_name,
                       out_dtype=str(y.dtype).replace("torch.", "")))

    for dt_name in ("float16", "bfloat16"):
        for bits in (4, 8):
            qmin, qmax = -2 ** (bits - 1), 2 ** (bits - 1) - 1
            thr = [float(rng.uniform(0.3, 5.0))]
            add("WeightsSymmetricInferableQuantizer", dict(num_bits=bits, threshold=thr, per_channel=False),
                adversarial((7, 33, 5), np.asarray(thr) / 2 ** (bits - 1), None, qmin, qmax), dt_name)
            for shape, axis in (((6, 40), 0), ((5, 7, 11), 1), ((1, 10, 10, 3), 3), ((3, 2048), 0), ((2, 3, 1032), 1)):
                C = shape[axis]
                thr = [float(v) for v in rng.uniform(0.05, 7.0, size=C)]
                add("WeightsSymmetricInferableQuantizer",
                    dict(num_bits=bits, threshold=thr, per_channel=True, channel_axis=axis),
                    adversarial(shape, np.asarray(thr) / 2 ** (bits - 1), axis, qmin, qmax), dt_name)
            thr = [float(2.0 ** e) for e in rng.integers(-3, 3, size=6)]
            add("WeightsPOTInferableQuantizer", dict(num_bits=bits, threshold=thr, per_channel=True, channel_axis=0),
                adversarial((6, 48), np.asarray(thr) / 2 ** (bits - 1), 0, qmin, qmax), dt_name)
            lo = [float(v) for v in rng.uniform(-4.0, 0.5, size=5)]
            hi = [float(a + d) for a, d in zip(lo, rng.uniform(0.2, 6.0, size=5))]
            kw = dict(num_bits=bits, min_range=lo, max_range=hi, per_channel=True, channel_axis=1)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                q = refq.WeightsUniformInferableQuantizer(**kw)
            add("WeightsUniformInferableQuantizer", kw,
                adversarial((3, 5, 24), q.scales.numpy().reshape(-1), 1, 0, 2 ** bits - 1, q.zero_points.numpy().reshape(-1)),
                dt_name)
            for signed in (True, False):
                thr = [float(rng.uniform(0.5, 6.0))]
                sc = np.asarray(thr) / (2 ** (bits - 1) if signed else 2 ** bits)
                dom = (qmin, qmax) if signed else (0, 2 ** bits - 1)
                add("ActivationSymmetricInferableQuantizer", dict(num_bits=bits, threshold=thr, signed=signed),
                    adversarial((2, 3, 17, 9), sc, None, *dom), dt_name)
            add("ActivationPOTInferableQuantizer", dict(num_bits=bits, threshold=[4.0], signed=True),
                adversarial((3, 50), np.asarray([4.0]) / 2 ** (bits - 1), None, qmin, qmax), dt_name)
            kw = dict(num_bits=bits, min_range=[-2.5], max_range=[3.1])
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                q = refq.ActivationUniformInferableQuantizer(**kw)
            add("ActivationUniformInferableQuantizer", kw,
                adversarial((2, 3, 12, 12), [q.scale], None, 0, 2 ** bits - 1, [q.zero_point]), dt_name)
        lut16 = list(workloads.CFG4_LUT)
        lut8 = [22.0, -53.0, 62.0, 0.0, -66.0, -21.0, 44.0, -40.0]
        for lut, bits in ((lut16, 4), (lut8, 3)):
            thr = [float(rng.uniform(0.5, 3.0))]
            add("WeightsLUTSymmetricInferableQuantizer",
                dict(num_bits=bits, lut_values=lut, threshold=thr, per_channel=False), lut_input((9, 31), thr, None), dt_name)
            for shape, axis in (((6, 37), 0), ((5, 7, 12), 1), ((3, 2048), 0)):
                C = shape[axis]
                thr = [float(v) for v in rng.uniform(0.05, 4.0, size=C)]
                add("WeightsLUTSymmetricInferableQuantizer",
                    dict(num_bits=bits, lut_values=lut, threshold=thr, per_channel=True, channel_axis=axis,
                         input_rank=len(shape)), lut_input(shape, thr, axis), dt_name)
            add("ActivationLutPOTInferableQuantizer", dict(num_bits=bits, lut_values=lut, threshold=[2.0], signed=True),
                lut_input((2, 3, 8, 8), [2.0], None), dt_name)
            add("ActivationLutPOTInferableQuantizer", dict(num_bits=bits, lut_values=lut, threshold=[0.5], signed=True),
                lut_input((3, 700), [0.5], None), dt_name)
        add("ActivationLutPOTInferableQuantizer",
            dict(num_bits=3, lut_values=[0.0, 13.0, 50.0, 90.0, 128.0, 200.0, 255.0, 256.0], threshold=[4.0], signed=False),
            np.abs(lut_input((3, 64), [4.0], None)) * np.float32(1.3), dt_name)
    np.savez_compressed(os.path.join(OUT, "cases_half.npz"), **ha)
    with open(os.path.join(OUT, "cases_half.json"), "w") as f:
        json.dump(dict(meta=dict(generator="tools/gen_golden.py", torch=torch.__version__), cases=hc), f, indent=1)
    print(len(hc), "half-precision cases")


def gen_half_bounds_cases():
    """Half-precision ACTIVATIONS through ActivationLutPOT with lut_values_bitwidth > 8: torch.clip converts the clip
    bounds to the tensor's type (quantizer_utils.py:129), so 2^k - 1 becomes 2^k in bfloat16 (and in float16 from 12 bits
    on) and 65535 does not fit float16 at all (RuntimeError).  Inputs sit around the top and bottom of the clip range;
    stored widened to float32 like cases_half."""
    hc, ha = [], {}
    g = np.random.default_rng(4242)
    for dt_name in ("float16", "bfloat16"):
        dt = getattr(torch, dt_name)
        for B, signed in ((9, False), (10, True), (10, False), (12, True), (12, False), (16, True), (16, False)):
            mult = float(2 ** (B - int(signed)))
            cmin, cmax = (float(-2 ** (B - 1)), float(2 ** (B - 1) - 1)) if signed else (0.0, float(2 ** B - 1))
            top = int(cmax)
            lo_c = int(cmin)
            # centres that make the two candidate clip bounds decide differently, plus a spread
            lut = sorted({top - 3, top - 1, top if signed else top + 1, lo_c, lo_c + 2, 0, top // 2, top // 3})
            lut = [float(v) for v in g.permutation(lut)]
            nb = int(np.ceil(np.log2(len(lut))))
            thr = [float(2.0 ** g.integers(-1, 3))]
            u = np.concatenate([g.uniform(-1.3, 1.3, 600), (cmax + g.integers(-6, 7, 200)) / mult,
                                (cmin + g.integers(-6, 7, 100)) / mult, [0.0, 1.0, -1.0, 0.999, 1.001]]).astype(np.float32)
            x32 = (u * np.float32(thr[0])).astype(np.float32).reshape(5, -1)
            xt = torch.from_numpy(np.ascontiguousarray(x32)).to(dt)
            kwargs = dict(num_bits=nb, lut_values=lut, threshold=thr, signed=signed, lut_values_bitwidth=B)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                q = refq.ActivationLutPOTInferableQuantizer(**kwargs)
            cid = f"hb{len(hc):03d}"
            rec = dict(id=cid, cls="ActivationLutPOTInferableQuantizer", kwargs=kwargs, shape=list(x32.shape), in_dtype=dt_name)
            ha[cid + "_x"] = xt.float().numpy()
            try:
                y = q(xt.clone())
                ha[cid + "_y"] = y.detach().float().numpy()
                rec["out_dtype"] = str(y.dtype).replace("torch.", "")
            except RuntimeError as e:
                rec["error"] = str(e)
            hc.append(rec)
    np.savez_compressed(os.path.join(OUT, "cases_half_bounds.npz"), **ha)
    meta = dict(reference="sony/mct_quantizers v%s" % ref.__version__, torch=torch.__version__, numpy=np.__version__,
                generator="tools/gen_golden.py --half-bounds-only")
    with open(os.path.join(OUT, "cases_half_bounds.json"), "w") as f:
        json.dump(dict(meta=meta, cases=hc), f, indent=1)
    print(f"{len(hc)} half-bounds cases ({sum('error' in c for c in hc)} raising), {sum(a.nbytes for a in ha.values())} array bytes")


def gen_f64_cases():
    """float64 inputs (the reference hands them to ATen unchanged).  ATen's double path is its own arithmetic
    (double product x * (double)(1.0f / s), per-channel results as double products, per-tensor ones as widened
    float32 products; LUT distances in double with a float32 result) -- these fixtures pin it.
    Inputs are float64 values that are NOT float32-representable, plus exact double ties."""
    dc, da = [], {}

    def widen(x32, scales, axis):
        x = x32.astype(np.float64)
        noise = rng.standard_normal(x.shape) * 2.0 ** -30
        x = x * (1.0 + noise)
        # exact ties of the DOUBLE product: x = (k + 0.5) / (double)(1.0f / s), nudged one double ulp either way
        s = np.asarray(scales, dtype=np.float32).reshape(-1)
        inv = (np.float32(1.0) / s).astype(np.float64)
        if axis is None:
            invb = np.broadcast_to(inv[0], x.shape)
        else:
            bs = [1] * x.ndim
            bs[axis] = -1
            invb = np.broadcast_to(inv.reshape(bs), x.shape)
        flat, invf = x.reshape(-1), invb.reshape(-1)
        pos = rng.choice(flat.size, size=max(3, flat.size // 6), replace=False)
        k = rng.integers(-140, 141, size=pos.size).astype(np.float64) + 0.5
        t = k / invf[pos]
        kind = rng.integers(0, 3, size=pos.size)
        t = np.where(kind == 1, np.nextafter(t, np.inf), t)
        t = np.where(kind == 2, np.nextafter(t, -np.inf), t)
        flat[pos] = t
        return flat.reshape(x.shape)

    def add(cls_name, kwargs, x64, memory_format=None):
        xt = torch.from_numpy(np.ascontiguousarray(x64))
        if memory_format == "channels_last":
            xt = xt.contiguous(memory_format=torch.channels_last)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            q = getattr(refq, cls_name)(**kwargs)
        y = q(xt.clone())
        cid = f"d{len(dc):03d}"
        da[cid + "_x"] = x64
        da[cid + "_y"] = y.detach().numpy().copy()
        dc.append(dict(id=cid, cls=cls_name, kwargs=kwargs, shape=list(x64.shape), in_dtype="float64",
                       out_dtype=str(y.dtype).replace("torch.", ""), memory_format=memory_format))

    for bits in (3, 8):
        qmin, qmax = -2 ** (bits - 1), 2 ** (bits - 1) - 1
        thr = [float(rng.uniform(0.3, 5.0))]
        sc = np.asarray(thr) / 2 ** (bits - 1)
        add("WeightsSymmetricInferableQuantizer", dict(num_bits=bits, threshold=thr, per_channel=False),
            widen(adversarial((7, 33, 5), sc, None, qmin, qmax), sc, None))
        for shape, axis, mf in (((6, 37), 0, None), ((5, 7, 11), 1, None), ((1, 10, 10, 3), 3, None), ((3, 2048), 0, None),
                                ((2, 6, 5, 4), 1, "channels_last")):
            C = shape[axis]
            thr = [float(v) for v in rng.uniform(0.05, 7.0, size=C)]
            sc = np.asarray(thr) / 2 ** (bits - 1)
            add("WeightsSymmetricInferableQuantizer",
                dict(num_bits=bits, threshold=thr, per_channel=True, channel_axis=axis),
                widen(adversarial(shape, sc, axis, qmin, qmax), sc, axis), mf)
        thr = [float(2.0 ** e) for e in rng.integers(-3, 3, size=6)]
        sc = np.asarray(thr) / 2 ** (bits - 1)
        add("WeightsPOTInferableQuantizer", dict(num_bits=bits, threshold=thr, per_channel=True, channel_axis=0),
            widen(adversarial((6, 48), sc, 0, qmin, qmax), sc, 0))
        lo = [float(v) for v in rng.uniform(-4.0, 0.5, size=5)]
        hi = [float(a + d) for a, d in zip(lo, rng.uniform(0.2, 6.0, size=5))]
        kw = dict(num_bits=bits, min_range=lo, max_range=hi, per_channel=True, channel_axis=1)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            q = refq.WeightsUniformInferableQuantizer(**kw)
        sc = q.scales.numpy().reshape(-1)
        add("WeightsUniformInferableQuantizer", kw,
            widen(adversarial((3, 5, 24), sc, 1, 0, 2 ** bits - 1, q.zero_points.numpy().reshape(-1)), sc, 1))
        kw = dict(num_bits=bits, min_range=[-1.3], max_range=[2.2], per_channel=False)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            q = refq.WeightsUniformInferableQuantizer(**kw)
        sc = q.scales.numpy().reshape(-1)
        add("WeightsUniformInferableQuantizer", kw,
            widen(adversarial((11, 23), sc, None, 0, 2 ** bits - 1, q.zero_points.numpy().reshape(-1)), sc, None))
        for signed in (True, False):
            thr = [float(rng.uniform(0.5, 6.0))]
            sc = np.asarray(thr) / (2 ** (bits - 1) if signed else 2 ** bits)
            dom = (qmin, qmax) if signed else (0, 2 ** bits - 1)
            add("ActivationSymmetricInferableQuantizer", dict(num_bits=bits, threshold=thr, signed=signed),
                widen(adversarial((2, 3, 17, 9), sc, None, *dom), sc, None))
        sc = np.asarray([4.0]) / 2 ** (bits - 1)
        add("ActivationPOTInferableQuantizer", dict(num_bits=bits, threshold=[4.0], signed=True),
            widen(adversarial((3, 50), sc, None, qmin, qmax), sc, None))
        kw = dict(num_bits=bits, min_range=[-2.5], max_range=[3.1])
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            q = refq.ActivationUniformInferableQuantizer(**kw)
        add("ActivationUniformInferableQuantizer", kw,
            widen(adversarial((2, 3, 12, 12), [q.scale], None, 0, 2 ** bits - 1, [q.zero_point]), [q.scale], None))

    def lut64(shape, thr, axis):
        x = lut_input(shape, thr, axis).astype(np.float64)
        return x * (1.0 + rng.standard_normal(x.shape) * 2.0 ** -30)

    lut16 = list(workloads.CFG4_LUT)
    lut8 = [22.0, -53.0, 62.0, 0.0, -66.0, -21.0, 44.0, -40.0]
    for lut, bits in ((lut16, 4), (lut8, 3)):
        thr = [float(rng.uniform(0.5, 3.0))]
        add("WeightsLUTSymmetricInferableQuantizer",
            dict(num_bits=bits, lut_values=lut, threshold=thr, per_channel=False), lut64((9, 31), thr, None))
        for shape, axis in (((6, 37), 0), ((5, 7, 12), 1), ((3, 2048), 0)):
            C = shape[axis]
            thr = [float(v) for v in rng.uniform(0.05, 4.0, size=C)]
            add("WeightsLUTSymmetricInferableQuantizer",
                dict(num_bits=bits, lut_values=lut, threshold=thr, per_channel=True, channel_axis=axis,
                     input_rank=len(shape)), lut64(shape, thr, axis))
        add("WeightsLUTPOTInferableQuantizer",
            dict(num_bits=bits, lut_values=lut, threshold=[2.0, 0.5, 1.0, 4.0], per_channel=True, channel_axis=0, input_rank=2),
            lut64((4, 300), [2.0, 0.5, 1.0, 4.0], 0))
        add("ActivationLutPOTInferableQuantizer", dict(num_bits=bits, lut_values=lut, threshold=[2.0], signed=True),
            lut64((2, 3, 8, 8), [2.0], None))
        add("ActivationLutPOTInferableQuantizer", dict(num_bits=bits, lut_values=lut, threshold=[0.5], signed=True, eps=1e-3),
            lut64((3, 700), [0.5], None))
    add("ActivationLutPOTInferableQuantizer",
        dict(num_bits=3, lut_values=[0.0, 13.0, 50.0, 90.0, 128.0, 200.0, 255.0, 256.0], threshold=[4.0], signed=False),
        np.abs(lut64((3, 64), [4.0], None)) * 1.3)
    np.savez_compressed(os.path.join(OUT, "cases_f64.npz"), **da)
    with open(os.path.join(OUT, "cases_f64.json"), "w") as f:
        json.dump(dict(meta=dict(generator="tools/gen_golden.py --f64-only", torch=torch.__version__,
                                 reference="sony/mct_quantizers v%s" % ref.__version__), cases=dc), f, indent=1)
    print(len(dc), "float64 cases")


def gen_extra_sha():
    """Digests of the reference outputs for the other batch sizes SURVEY §8(d) names for config 3 (N = 1, 64, 256);
    merged into full_sha.json next to the N = 8 entry."""
    path = os.path.join(OUT, "full_sha.json")
    with open(path) as f:
        doc = json.load(f)
    torch.set_num_threads(os.cpu_count() or 1)
    for n in (1, 64, 256):
        x = workloads.make_input("cfg3", batch=n)
        wl = workloads.make_workload("cfg3", x)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            q = getattr(refq, wl.quantizer)(**wl.kwargs)
        y = q(torch.from_numpy(x)).numpy()
        doc["configs"][f"cfg3_n{n}"] = dict(shape=list(x.shape), quantizer=wl.quantizer, batch=n,
                                            x_sha256=hashlib.sha256(x.tobytes()).hexdigest(),
                                            y_sha256=hashlib.sha256(np.ascontiguousarray(y).tobytes()).hexdigest(),
                                            y_sum_f64=float(np.sum(y, dtype=np.float64)))
        print(f"cfg3 N={n}", doc["configs"][f"cfg3_n{n}"]["y_sha256"], flush=True)
    with open(path, "w") as f:
        json.dump(doc, f, indent=1)


def gen_shard_sha():
    """Per-rank digests of BASELINE config 5 sharded by dim 0: the REFERENCE quantizes the whole tensor,
    its output is cut into the contiguous row blocks of every world size 1..8 (block = ceil(rows / world), the rule of
    mct_quantizers_amd/sharded.py:row_block, restated here so that the package under test is not imported) and each
    block is hashed.  bench.py's N > 1 line checks every rank's shard against these (bench_dist.shard_digest).  Also the
    small shapes of the CPU dry run / gloo tests."""
    out = {}
    torch.set_num_threads(os.cpu_count() or 1)
    for rows, cols in ((8192, 8192), (64, 128), (32, 128), (13, 128)):
        x = workloads.make_input("cfg5", shape=(rows, cols))
        wl = workloads.make_workload("cfg5", x)
        y = np.ascontiguousarray(getattr(refq, wl.quantizer)(**wl.kwargs)(torch.from_numpy(x)).numpy())
        rec = dict(quantizer=wl.quantizer, y_sha256=hashlib.sha256(y.tobytes()).hexdigest(), shards={})
        for world in range(1, 9):
            per = -(-rows // world)
            rec["shards"][str(world)] = [
                hashlib.sha256(y[min(rows, r * per):min(rows, (r + 1) * per)].tobytes()).hexdigest() for r in range(world)]
        out[f"{rows}x{cols}"] = rec
        print(f"cfg5 {rows}x{cols}", rec["y_sha256"], flush=True)
    meta = dict(reference="sony/mct_quantizers v%s" % ref.__version__, torch=torch.__version__, numpy=np.__version__,
                generator="tools/gen_golden.py --shard-sha-only", rule="rank r owns rows [min(R, r*ceil(R/N)), min(R, (r+1)*ceil(R/N)))")
    with open(os.path.join(OUT, "shard_sha.json"), "w") as f:
        json.dump(dict(meta=meta, cfg5=out), f, indent=1)


def gen_traced_wrapper_pickle():
    """An fx-traced REFERENCE wrapper whose weights quantizer is per-tensor: its graph holds
    torch.fake_quantize_per_tensor_affine with TENSOR scale / zero point (graph values), the saved-model form of
    weights_symmetric_inferable_quantizer.py:147-151; plus a per-channel one.  Data only (class paths + tensors +
    fx's generated forward)."""
    import torch.nn as nn
    torch.manual_seed(11)

    class Leafy(torch.fx.Tracer):
        def is_leaf_module(self, m, qualname):
            return isinstance(m, (nn.Linear, nn.Conv2d)) or super().is_leaf_module(m, qualname)

    lin = nn.Linear(12, 5)
    lin2 = nn.Linear(5, 4)
    thr2 = [float(v) for v in lin2.weight.detach().abs().amax(dim=1)]
    net = nn.Sequential()
    net.add_module("l1", ref.PytorchQuantizationWrapper(
        lin, {"weight": refq.WeightsSymmetricInferableQuantizer(8, [float(lin.weight.detach().abs().max())], False),
              "bias": refq.WeightsUniformInferableQuantizer(8, [-0.6], [0.7], False)}))
    net.add_module("a1", ref.PytorchActivationQuantizationHolder(refq.ActivationSymmetricInferableQuantizer(8, [2.0], True)))
    net.add_module("l2", ref.PytorchQuantizationWrapper(
        lin2, {"weight": refq.WeightsSymmetricInferableQuantizer(4, thr2, True, 0)}))
    x = torch.randn(6, 12)
    y = net(x)
    graph = Leafy().trace(net)
    traced = torch.fx.GraphModule(net, graph)
    yt = traced(x)
    assert torch.equal(y, yt)
    kinds = [(n.op, str(n.target)) for n in traced.graph.nodes]
    # what the quantizers inside produce on the reference, for bit-exact per-quantizer checks
    w1 = net.l1.weights_quantizers["weight"](lin.weight.detach().clone())
    b1 = net.l1.weights_quantizers["bias"](lin.bias.detach().clone())
    w2 = net.l2.weights_quantizers["weight"](lin2.weight.detach().clone())
    torch.save(traced, os.path.join(OUT, "ref_traced_wrapper.pth"))
    np.savez_compressed(os.path.join(OUT, "ref_traced_wrapper_io.npz"), x=x.numpy(), y=y.detach().numpy(),
                        w1=w1.detach().numpy(), b1=b1.detach().numpy(), w2=w2.detach().numpy())
    with open(os.path.join(OUT, "ref_traced_wrapper_nodes.json"), "w") as f:
        json.dump(kinds, f, indent=1)
    print("traced reference wrapper:", [k for k in kinds if "fake_quantize" in k[1]])

    # The reference's weights quantizers traced with the WEIGHT as a graph input: symbolic_trace records
    # torch.fake_quantize_per_tensor_affine(w, <tensor scale>, <tensor zero point>, ...) -- the tensor-qparams
    # overload, its qparams lifted to _tensor_constant* attributes -- and fake_quantize_per_channel_affine.
    class WQ(nn.Module):
        def __init__(self):
            super().__init__()
            self.q1 = refq.WeightsSymmetricInferableQuantizer(8, [1.7], False)
            self.q2 = refq.WeightsSymmetricInferableQuantizer(4, [1.0, 2.0, 0.7], True, 0)
            self.q3 = refq.WeightsUniformInferableQuantizer(8, [-0.6], [0.7], False)

        def forward(self, w, v):
            return self.q1(w), self.q3(w), self.q2(v)

    tq = torch.fx.symbolic_trace(WQ())
    w = torch.randn(9, 33) * 1.2
    v = torch.randn(3, 257)
    o1, o3, o2 = tq(w, v)
    torch.save(tq, os.path.join(OUT, "ref_traced_weight_quantizers.pth"))
    np.savez_compressed(os.path.join(OUT, "ref_traced_weight_quantizers_io.npz"), w=w.numpy(), v=v.numpy(),
                        o1=o1.numpy(), o3=o3.numpy(), o2=o2.numpy())
    print("traced reference weights quantizers:", [str(n.target) for n in tq.graph.nodes if n.op == "call_function"])


def gen_pickled_reference_models():
    """Pickles of REFERENCE objects (class paths + state only, no source) for the unpickle-compat tests:
    a small module tree built from the reference's wrapper/holders/quantizers, and an fx-traced holder."""
    import torch.nn as nn
    torch.manual_seed(7)

    conv = nn.Conv2d(3, 4, 3)
    lin = nn.Linear(4, 3)
    thr_c = [float(v) for v in conv.weight.detach().abs().amax(dim=(1, 2, 3))]
    net = nn.Sequential()
    net.add_module("conv", ref.PytorchQuantizationWrapper(
        conv, {"weight": refq.WeightsSymmetricInferableQuantizer(8, thr_c, True, 0)}))
    net.add_module("act", ref.PytorchActivationQuantizationHolder(
        refq.ActivationUniformInferableQuantizer(8, [-1.0], [3.0])))
    net.add_module("pool", nn.AdaptiveAvgPool2d(1))
    net.add_module("flat", nn.Flatten())
    net.add_module("fln", ref.PytorchFLNActivationQuantizationHolder(
        refq.ActivationPOTInferableQuantizer(4, [2.0], True)))
    net.add_module("lin", ref.PytorchQuantizationWrapper(
        lin, {"weight": refq.WeightsLUTSymmetricInferableQuantizer(
            3, [-100.0, -50.0, -10.0, 0.0, 10.0, 50.0, 100.0, 127.0], [1.0], False),
            "bias": refq.WeightsUniformInferableQuantizer(8, [-1.0], [1.0], False)}))
    net.add_module("keep", ref.PytorchPreservingActivationQuantizationHolder(
        refq.ActivationLutPOTInferableQuantizer(2, [-100.0, 0.0, 60.0, 127.0], [4.0], True), quantization_bypass=False))
    x = torch.randn(2, 3, 10, 10)
    y = net(x)
    torch.save(net, os.path.join(OUT, "ref_model.pth"))
    traced = torch.fx.symbolic_trace(ref.PytorchActivationQuantizationHolder(
        refq.ActivationUniformInferableQuantizer(3, [-2.0], [2.0])))
    yt = traced(x)
    torch.save(traced, os.path.join(OUT, "ref_traced_holder.pth"))
    np.savez_compressed(os.path.join(OUT, "ref_model_io.npz"), x=x.numpy(), y=y.detach().numpy(), y_traced=yt.numpy())
    print("pickled reference model:", os.path.getsize(os.path.join(OUT, "ref_model.pth")), "bytes")


# ------------------------------------------------------------------------------------------
# ONNX-export branch (enable_custom_impl + tracing): outputs of the export-time arithmetic and the
# nodes the symbolic functions emit
# ------------------------------------------------------------------------------------------

def traced_call(q, xt):
    """q(xt) evaluated while torch.jit is tracing (the reference's switch into its export arithmetic)."""
    box = {}

    def fn(t):
        box["y"] = q(t)
        return box["y"]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        torch.jit.trace(fn, xt, check_trace=False)
    return box["y"].detach()


def onnx_nodes(module, example):
    """[(kind, {attr: value})] of the ONNX graph the TorchScript exporter builds (no `onnx` package needed)."""
    from torch.onnx._internal.torchscript_exporter import utils
    from torch.onnx._internal.torchscript_exporter._globals import GLOBALS
    GLOBALS.export_onnx_opset_version = 16
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        with utils.exporter_context(module, torch.onnx.TrainingMode.EVAL, False):
            graph, _, _ = utils._model_to_graph(module, (example,), do_constant_folding=False)
    out = []
    for n in graph.nodes():
        attrs = {}
        for a in n.attributeNames():
            kind = n.kindOf(a)
            attrs[a] = dict(i=n.i, f=n.f, s=n.s, t=lambda k: n.t(k).tolist(), is_=n.is_, fs=n.fs)[
                "is_" if kind == "is" else kind](a)
        out.append([n.kind(), attrs, len(list(n.inputs()))])
    return out


def export_configs():
    """(name, class, kwargs, input shape, holder?) for every quantizer with an export branch."""
    er = np.random.default_rng(20251004)
    u = lambda lo, hi, n: [float(v) for v in er.uniform(lo, hi, size=n)]   # noqa: E731
    lut16 = [-128., -96., -64., -40., -24., -12., -5., 0., 5., 12., 24., 40., 64., 96., 120., 127.]
    cfgs = []
    for bits in (2, 4, 8):
        cfgs += [
            ("wsym_pc0", "WeightsSymmetricInferableQuantizer", dict(num_bits=bits, threshold=u(0.05, 7, 6), per_channel=True, channel_axis=0), (6, 37)),
            ("wsym_pc1", "WeightsSymmetricInferableQuantizer", dict(num_bits=bits, threshold=u(0.05, 7, 7), per_channel=True, channel_axis=1), (5, 7, 11)),
            ("wsym_pclast", "WeightsSymmetricInferableQuantizer", dict(num_bits=bits, threshold=u(0.05, 7, 8), per_channel=True, channel_axis=-1), (3, 9, 8)),
            ("wsym_pt", "WeightsSymmetricInferableQuantizer", dict(num_bits=bits, threshold=u(0.3, 5, 1), per_channel=False), (7, 33, 5)),
            ("wpot_pc", "WeightsPOTInferableQuantizer", dict(num_bits=bits, threshold=[float(2.0 ** e) for e in er.integers(-4, 4, size=6)], per_channel=True, channel_axis=0), (6, 40)),
            ("wpot_pt", "WeightsPOTInferableQuantizer", dict(num_bits=bits, threshold=[2.0], per_channel=False), (9, 9)),
            ("wuni_pc", "WeightsUniformInferableQuantizer", dict(num_bits=bits, min_range=u(-4, -0.1, 6), max_range=u(0.2, 6, 6), per_channel=True, channel_axis=0), (6, 37)),
            ("wuni_pc_mixed", "WeightsUniformInferableQuantizer", dict(num_bits=bits, min_range=[-1.0, 0.5, -3.0, -0.2], max_range=[2.0, 4.0, -1.0, 0.7], per_channel=True, channel_axis=1), (5, 4, 9)),
            ("wuni_pt", "WeightsUniformInferableQuantizer", dict(num_bits=bits, min_range=[-1.3], max_range=[2.9], per_channel=False), (11, 13)),
            ("asym_s", "ActivationSymmetricInferableQuantizer", dict(num_bits=bits, threshold=u(0.5, 6, 1), signed=True), (2, 3, 17, 9)),
            ("asym_u", "ActivationSymmetricInferableQuantizer", dict(num_bits=bits, threshold=u(0.5, 6, 1), signed=False), (2, 3, 17, 9)),
            ("apot_s", "ActivationPOTInferableQuantizer", dict(num_bits=bits, threshold=[4.0], signed=True), (3, 50)),
            ("apot_u", "ActivationPOTInferableQuantizer", dict(num_bits=bits, threshold=[0.5], signed=False), (3, 50)),
        ]
        for lo, hi in ((-2.5, 3.1), (3.0, 10.0), (-7.0, -1.0), (-0.37, 0.91)):
            cfgs.append((f"auni_{lo}_{hi}", "ActivationUniformInferableQuantizer", dict(num_bits=bits, min_range=[lo], max_range=[hi]), (2, 3, 12, 12)))
    cfgs += [
        ("wlut_pc", "WeightsLUTSymmetricInferableQuantizer", dict(num_bits=4, lut_values=lut16, threshold=u(0.5, 3, 6), per_channel=True, channel_axis=0, input_rank=2), (6, 50)),
        ("wlut_pt", "WeightsLUTSymmetricInferableQuantizer", dict(num_bits=3, lut_values=[22., -53., 62., 0., -66., -21., 44., -40.], threshold=[1.7], per_channel=False), (8, 31)),
        ("wlutpot_pc", "WeightsLUTPOTInferableQuantizer", dict(num_bits=4, lut_values=lut16, threshold=[0.5, 1.0, 2.0, 4.0], per_channel=True, channel_axis=1, input_rank=3), (3, 4, 20)),
        ("wlutpot_pt", "WeightsLUTPOTInferableQuantizer", dict(num_bits=2, lut_values=[-25., 25., 0., 100.], threshold=[2.0], per_channel=False), (8, 31)),
    ]
    return cfgs


def export_input(er, q, cls_name, kwargs, shape):
    """Ties, clip edges and values beyond them on the export grid of each channel."""
    n = int(np.prod(shape))
    if "LUT" in cls_name:
        thr = np.asarray(kwargs["threshold"], dtype=np.float32)
        x = er.uniform(-1.6, 1.6, size=n).astype(np.float32).reshape(shape)
        if kwargs.get("per_channel"):
            bs = [1] * len(shape)
            bs[kwargs["channel_axis"]] = -1
            x = x * thr.reshape(bs)
        else:
            x = x * thr[0]
        return np.ascontiguousarray(x.astype(np.float32))
    bits = kwargs["num_bits"]
    if "threshold" in kwargs:
        thr = np.asarray(kwargs["threshold"], dtype=np.float64)
        signed = kwargs.get("signed", True)
        step = thr / (2 ** (bits - 1) if signed else 2 ** bits)
        lo = -thr if signed else np.zeros_like(thr)
    else:
        lo = np.asarray(kwargs["min_range"], dtype=np.float64)
        hi = np.asarray(kwargs["max_range"], dtype=np.float64)
        step = (hi - lo) / (2 ** bits - 1)
    axis = kwargs.get("channel_axis") if kwargs.get("per_channel") else None
    if axis is None:
        sb, lb = np.broadcast_to(step[0], shape), np.broadcast_to(lo[0], shape)
    else:
        bs = [1] * len(shape)
        bs[axis] = -1
        sb, lb = np.broadcast_to(step.reshape(bs), shape), np.broadcast_to(lo.reshape(bs), shape)
    k = er.integers(-3, 2 ** bits + 3, size=shape).astype(np.float64)
    frac = er.choice([0.0, 0.5, 0.5, 0.49999, 0.50001, 0.25], size=shape)
    x = (lb + (k + frac) * sb)
    x32 = x.astype(np.float32)
    bump = er.integers(-2, 3, size=shape)                      # neighbours of the tie, in ulps
    x32 = (x32.view(np.int32) + bump.astype(np.int32)).view(np.float32)
    flat = x32.reshape(-1)
    specials = np.asarray([0.0, -0.0, 1e-30, -1e-30, 1e30, -1e30, np.inf, -np.inf, np.nan], dtype=np.float32)
    if flat.size >= 4 * specials.size:
        flat[er.choice(flat.size, size=specials.size, replace=False)] = specials
    return np.ascontiguousarray(flat.reshape(shape))


def gen_export():
    er = np.random.default_rng(20251005)
    ecases, earrays, nodes = [], {}, []
    for name, cls_name, kwargs, shape in export_configs():
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            q = getattr(refq, cls_name)(**kwargs)
        q.enable_custom_impl()
        x = export_input(er, q, cls_name, kwargs, shape)
        y = traced_call(q, torch.from_numpy(x.copy()))
        cid = f"e{len(ecases):03d}"
        earrays[cid + "_x"], earrays[cid + "_y"] = x, y.numpy().copy()
        ecases.append(dict(id=cid, name=name, cls=cls_name, kwargs=kwargs, shape=list(shape)))
        # the node(s) this quantizer exports as
        if cls_name.startswith("Weights"):
            class Holder(torch.nn.Module):
                def __init__(self, quant, shp):
                    super().__init__()
                    self.q = quant
                    self.w = torch.nn.Parameter(torch.zeros(shp))

                def forward(self, t):
                    return t + self.q(self.w).sum()
            mod, ex = Holder(q, shape), torch.zeros(3)
        else:
            mod, ex = ref.PytorchActivationQuantizationHolder(q), torch.zeros(shape)
        nodes.append(dict(id=cid, nodes=[n for n in onnx_nodes(mod, ex) if n[0].startswith("mct_quantizers::") or n[0] == "onnx::Constant"]))
    np.savez_compressed(os.path.join(OUT, "export_cases.npz"), **earrays)
    meta = dict(reference="sony/mct_quantizers v%s" % ref.__version__, torch=torch.__version__,
                generator="tools/gen_golden.py --export-only")
    with open(os.path.join(OUT, "export_cases.json"), "w") as f:
        json.dump(dict(meta=meta, cases=ecases, onnx_nodes=nodes), f, indent=1)
    print(f"{len(ecases)} export cases, {sum(a.nbytes for a in earrays.values())} array bytes")


if __name__ == "__main__":
    if "--pickles-only" in sys.argv:
        os.makedirs(OUT, exist_ok=True)
        gen_pickled_reference_models()
    elif "--half-bounds-only" in sys.argv:
        gen_half_bounds_cases()
    elif "--half-only" in sys.argv:
        gen_half_cases()
    elif "--export-only" in sys.argv:
        gen_export()
    elif "--f64-only" in sys.argv:
        gen_f64_cases()
    elif "--extra-sha-only" in sys.argv:
        gen_extra_sha()
    elif "--traced-wrapper-only" in sys.argv:
        gen_traced_wrapper_pickle()
    elif "--shard-sha-only" in sys.argv:
        gen_shard_sha()
    else:
        main()
        gen_half_cases()
        gen_pickled_reference_models()
        gen_export()
        gen_f64_cases()
        gen_extra_sha()
        gen_traced_wrapper_pickle()
        gen_shard_sha()
