"""Parity of the gfx950 kernels with the oracle / the reference goldens.  Needs an MI355X: -m gpu.

Bar: bit-exact float32 outputs (hence bit-exact integer clamp indices and LUT indices) on finite
inputs with |x/scale| < 2**31.  Everything here reaches the kernels through the C ABI
(ctypes -> libmctq_hip.so), either directly or via the quantizer classes.
"""
import hashlib
import warnings

import numpy as np
import pytest
import torch

from conftest import bits_equal, finite_equal, first_mismatch, load_json

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    from mct_quantizers_amd.hip import native
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    return native.load()          # raises if the .so is missing: no silent fallback


def _dev(a: np.ndarray) -> torch.Tensor:
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _stream():
    return torch.cuda.current_stream().cuda_stream


# ---------------------------------------------------------------------------------------------
# 1. golden fixtures (generated from the reference) through the public quantizer classes
# ---------------------------------------------------------------------------------------------

def test_golden_cases_via_quantizer_classes(lib, golden_cases):
    import mct_quantizers_amd as mq
    Q = mq.pytorch_quantizers
    cases, arrays = golden_cases
    for c in cases:
        x_np = arrays[c["id"] + "_x"]
        want = arrays[c["id"] + "_y"]
        x = _dev(x_np)
        if c["memory_format"] == "channels_last":
            x = x.contiguous(memory_format=torch.channels_last)
        elif c["memory_format"] == "transposed":
            x = x.transpose(0, -1).contiguous().transpose(0, -1)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            q = getattr(Q, c["cls"])(**c["kwargs"])
        y = q(x)
        assert y.is_cuda and y.shape == x.shape and y.dtype == torch.float32
        if "LUT" not in c["cls"] and "Lut" not in c["cls"]:
            assert y.stride() == x.stride(), c["id"]           # ATen keeps the input's strides
        else:
            assert y.is_contiguous(), c["id"]                  # the LUT chain ends in a gather: contiguous result
        got = y.cpu().numpy()
        assert bits_equal(got, want), f'{c["id"]} {c["cls"]} {c["shape"]} {c["memory_format"]}: ' \
                                      f'{first_mismatch(got, want, x_np)}'


def test_half_precision_golden_cases_via_quantizer_classes(lib, half_cases):
    """float16 / bfloat16 tensors: affine quantizers keep the type, LUT quantizers return float32."""
    import mct_quantizers_amd as mq
    Q = mq.pytorch_quantizers
    cases, arrays = half_cases
    for c in cases:
        x32 = arrays[c["id"] + "_x"]
        x = _dev(x32).to(getattr(torch, c["in_dtype"]))
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            q = getattr(Q, c["cls"])(**c["kwargs"])
        y = q(x)
        assert y.is_cuda and str(y.dtype) == "torch." + c["out_dtype"], c["id"]
        got = y.float().cpu().numpy()
        want = arrays[c["id"] + "_y"]
        assert finite_equal(got, want, x32), f'{c["id"]} {c["cls"]} {c["in_dtype"]} {c["shape"]}: ' \
                                             f'{first_mismatch(got, want, x32)}'


@pytest.mark.parametrize("dt", ["float16", "bfloat16"])
@pytest.mark.parametrize("outer,C,inner,offset", [(1, 1, 5, 0), (7, 3, 1, 0), (33, 5, 2, 1), (2, 16, 8, 0), (3, 64, 12, 3),
                                                  (1, 300, 576, 0), (2, 6, 1024, 0), (3, 5, 1032, 0), (1, 64, 4096, 0),
                                                  (2, 3, 11008, 8), (1, 20000, 1, 0), (1, 2, 70000, 0), (1, 1, 4099, 1),
                                                  (37, 64, 1, 0), (5, 4096, 1, 0), (9, 24, 1, 8)])
def test_abi_half_per_channel_and_per_tensor_vs_oracle(lib, dt, outer, C, inner, offset):
    from oracle import mctq_oracle as O
    code = {"float16": 1, "bfloat16": 2}[dt]
    tdt = getattr(torch, dt)
    rng = np.random.default_rng(C * 7 + inner + offset)
    qmin, qmax = -128, 127
    scales = rng.uniform(0.01, 0.2, size=C).astype(np.float32)
    zps = rng.integers(-5, 6, size=C).astype(np.int32)
    shape = (outer, C, inner)
    x32 = _tie_heavy(rng, shape, scales.reshape(1, C, 1), zps.reshape(1, C, 1).astype(np.float32), qmin, qmax)
    xh = _dev(x32.reshape(-1)).to(tdt)
    x_np = xh.float().cpu().numpy().reshape(shape)                 # the values actually stored
    n = x_np.size
    xb = torch.empty(n + offset, dtype=tdt, device="cuda")
    xb[offset:] = xh
    yb = torch.full_like(xb, 512.0)
    s_d, z_d = _dev(scales), _dev(zps)
    rc = lib.mctq_fq_per_channel(xb[offset:].data_ptr(), yb[offset:].data_ptr(), outer, C, inner, code,
                                 s_d.data_ptr(), z_d.data_ptr(), qmin, qmax, _stream())
    assert rc == 0, lib.mctq_last_error()
    want = O.narrow(O.fake_quant_affine(x_np, scales, zps, qmin, qmax, axis=1), dt)
    got = yb.float().cpu().numpy()
    assert finite_equal(got[offset:].reshape(shape), want, x_np), first_mismatch(got[offset:], want, x_np)
    if offset:
        assert np.all(got[:offset] == 512.0)
    yb.fill_(512.0)
    rc = lib.mctq_fq_per_tensor(xb[offset:].data_ptr(), yb[offset:].data_ptr(), n, code, float(scales[0]), int(zps[0]),
                                qmin, qmax, _stream())
    assert rc == 0, lib.mctq_last_error()
    want = O.narrow(O.fake_quant_affine(x_np, scales[0], zps[0], qmin, qmax), dt)
    got = yb.float().cpu().numpy()
    assert finite_equal(got[offset:].reshape(shape), want, x_np), first_mismatch(got[offset:], want, x_np)


# ---------------------------------------------------------------------------------------------
# 2. raw C-ABI calls vs the oracle: every launch shape, alignment and tail
# ---------------------------------------------------------------------------------------------

def _tie_heavy(rng, shape, s_b, zp_b, qmin, qmax):
    """Inputs dense in round-half ties and clamp edges of their own channel's grid."""
    n = int(np.prod(shape))
    x = (rng.standard_normal(n).astype(np.float32).reshape(shape) * s_b * np.float32(0.4 * (qmax - qmin)))
    k = rng.integers(qmin - 2, qmax + 3, size=shape).astype(np.float32) - zp_b
    kind = rng.integers(0, 6, size=shape)
    x = np.where(kind == 0, (k + np.float32(0.5)) * s_b, x)
    x = np.where(kind == 1, k * s_b, x)
    return x.astype(np.float32)


@pytest.mark.parametrize("n", [0, 1, 3, 4, 5, 1023, 1024, 4096 + 7, 256 * 4 * 4 * 3 + 1, 1 << 20])
@pytest.mark.parametrize("offset", [0, 1])
def test_abi_per_tensor_vs_oracle(lib, n, offset):
    from oracle import mctq_oracle as O
    rng = np.random.default_rng(n * 2 + offset)
    scale, zp, qmin, qmax = np.float32(0.0371), 17, 0, 255
    x_np = _tie_heavy(rng, (n + offset,), scale, np.float32(zp), qmin, qmax)
    xb = _dev(x_np)
    yb = torch.full_like(xb, 777.0)
    x = xb[offset:]
    y = yb[offset:]
    rc = lib.mctq_fq_per_tensor_f32(x.data_ptr(), y.data_ptr(), n, float(scale), zp, qmin, qmax, _stream())
    assert rc == 0, lib.mctq_last_error()
    want = O.fake_quant_affine(x_np[offset:], scale, zp, qmin, qmax)
    got = yb.cpu().numpy()
    assert bits_equal(got[offset:], want), first_mismatch(got[offset:], want, x_np[offset:])
    if offset:
        assert got[0] == 777.0                                   # nothing written outside [offset, offset+n)


CHANNEL_SHAPES = [
    # (outer, C, inner)
    (1, 1, 1), (1, 3, 1), (7, 3, 1), (1000, 3, 1), (33, 5, 2), (9, 4, 3), (2, 16, 4), (5, 7, 5),
    (3, 64, 12), (2, 8, 100), (1, 300, 576), (4, 6, 1020), (2, 6, 1024), (3, 5, 1028), (1, 64, 4096),
    (2, 3, 11008), (1, 5000, 7), (1, 20000, 1), (3, 1, 5000), (1, 2, 70000), (1, 1, 4099),
    (37, 64, 1), (5, 4096, 1), (1, 8, 1), (300, 12, 1),          # channel-last: the lastaxis shape (C % 4 == 0)
]


@pytest.mark.parametrize("outer,C,inner", CHANNEL_SHAPES)
@pytest.mark.parametrize("offset", [0, 3])
def test_abi_per_channel_vs_oracle(lib, outer, C, inner, offset):
    from oracle import mctq_oracle as O
    rng = np.random.default_rng(outer * 1315423911 % 100003 + C * 7 + inner + offset)
    bits = int(rng.choice([2, 4, 8]))
    qmin, qmax = 0, 2 ** bits - 1
    scales = rng.uniform(0.01, 0.9, size=C).astype(np.float32)
    zps = rng.integers(qmin, qmax + 1, size=C).astype(np.int32)
    shape = (outer, C, inner)
    x_np = _tie_heavy(rng, shape, scales.reshape(1, C, 1), zps.reshape(1, C, 1).astype(np.float32), qmin, qmax)
    n = x_np.size
    pad = 64                                                     # sentinels after the end, too
    xb = torch.zeros(n + offset + pad, dtype=torch.float32, device="cuda")
    xb[offset:offset + n] = _dev(x_np.reshape(-1))
    yb = torch.full_like(xb, 777.0)
    s_d, z_d = _dev(scales), _dev(zps)
    rc = lib.mctq_fq_per_channel_f32(xb[offset:].data_ptr(), yb[offset:].data_ptr(), outer, C, inner,
                                     s_d.data_ptr(), z_d.data_ptr(), qmin, qmax, _stream())
    assert rc == 0, lib.mctq_last_error()
    want, q_want = O.fake_quant_affine(x_np, scales, zps, qmin, qmax, axis=1, return_index=True)
    got_all = yb.cpu().numpy()
    assert np.all(got_all[:offset] == 777.0) and np.all(got_all[offset + n:] == 777.0)   # nothing outside [0, n)
    got = got_all[:offset + n]
    assert bits_equal(got[offset:].reshape(shape), want), first_mismatch(got[offset:], want, x_np)
    if offset:
        assert np.all(got[:offset] == 777.0)
    # integer clamp index recovered from the output equals the oracle's index
    q_got = np.rint(got[offset:].reshape(shape) / scales.reshape(1, C, 1)).astype(np.int64) + zps.reshape(1, C, 1)
    assert np.array_equal(q_got, q_want)


LUTS = {
    "l2": [-5.0, 5.0],
    "l3dup": [3.0, 3.0, -8.0],
    "l8": [22.0, -53.0, 62.0, 0.0, -66.0, -21.0, 44.0, -40.0],
    "l16": [-128.0, -96.0, -64.0, -40.0, -24.0, -12.0, -5.0, 0.0, 5.0, 12.0, 24.0, 40.0, 64.0, 96.0, 120.0, 127.0],
    "l40": [float(v) for v in np.random.default_rng(5).permutation(np.arange(-128, 128))[:40]],
    "l256": [float(v) for v in np.random.default_rng(6).permutation(np.arange(-128, 128))],
}


def _lut_inputs(rng, shape, thr_b):
    x = rng.standard_normal(int(np.prod(shape))).astype(np.float32).reshape(shape) * thr_b * np.float32(0.6)
    mid = (rng.integers(-130, 130, size=shape).astype(np.float32) + np.float32(0.5)) / np.float32(128.0) * thr_b
    x = np.where(rng.integers(0, 4, size=shape) == 0, mid, x).astype(np.float32)
    flat = x.reshape(-1)
    if flat.size >= 8:
        flat[:8] = np.asarray([0.0, -0.0, 1e-9, -1e-9, 1e-39, 3e5, -3e5, 1e30], dtype=np.float32)
    return x


@pytest.mark.parametrize("lut_name", list(LUTS))
@pytest.mark.parametrize("n,offset", [(0, 0), (5, 0), (4096 + 3, 0), (100000, 1)])
def test_abi_lut_per_tensor_vs_oracle(lib, lut_name, n, offset):
    from oracle import mctq_oracle as O
    rng = np.random.default_rng(n + len(lut_name))
    lut = LUTS[lut_name]
    thr = 2.0
    x_np = _lut_inputs(rng, (n + offset,), np.float32(thr))
    xb, lut_d = _dev(x_np), _dev(np.asarray(lut, dtype=np.float32))
    yb = torch.full_like(xb, 777.0)
    thr_div = float(np.float32(thr + 1e-8))
    rc = lib.mctq_lut_per_tensor_f32(xb[offset:].data_ptr(), yb[offset:].data_ptr(), n, thr_div, thr,
                                     lut_d.data_ptr(), len(lut), 128.0, -128.0, 127.0, _stream())
    assert rc == 0, lib.mctq_last_error()
    want = O.lut_quantize(x_np[offset:], lut, thr, True, 8, 1e-8)
    got = yb.cpu().numpy()
    assert bits_equal(got[offset:], want), first_mismatch(got[offset:], want, x_np[offset:])


@pytest.mark.parametrize("lut_name", ["l3dup", "l16", "l40", "l256"])
@pytest.mark.parametrize("outer,C,inner", [(1, 3, 1), (50, 3, 1), (4, 6, 5), (2, 8, 100), (2, 6, 1024), (3, 5, 1028),
                                           (1, 16, 11008), (1, 3000, 3), (41, 64, 1), (3, 4096, 1)])
def test_abi_lut_per_channel_vs_oracle(lib, lut_name, outer, C, inner):
    from oracle import mctq_oracle as O
    rng = np.random.default_rng(C * 31 + inner)
    lut = LUTS[lut_name]
    thr = rng.uniform(0.05, 4.0, size=C).astype(np.float32)
    shape = (outer, C, inner)
    x_np = _lut_inputs(rng, shape, thr.reshape(1, C, 1))
    x, t_d, lut_d = _dev(x_np), _dev(thr), _dev(np.asarray(lut, dtype=np.float32))
    y = torch.empty_like(x)
    rc = lib.mctq_lut_per_channel_f32(x.data_ptr(), y.data_ptr(), outer, C, inner, t_d.data_ptr(), 1e-8,
                                      lut_d.data_ptr(), len(lut), 128.0, -128.0, 127.0, _stream())
    assert rc == 0, lib.mctq_last_error()
    want, idx_want = O.lut_quantize(x_np, lut, thr, True, 8, 1e-8, per_channel=True, channel_axis=1,
                                    return_index=True)
    got = y.cpu().numpy()
    assert bits_equal(got, want), first_mismatch(got, want, x_np)
    # LUT index parity: the chosen centre is lut[idx_want] for every element
    centres = np.asarray(lut, dtype=np.float32)[idx_want] / np.float32(128.0) * thr.reshape(1, C, 1)
    assert bits_equal(got, centres.astype(np.float32))


@pytest.mark.parametrize("scale,zp,qmin,qmax", [(0.0371, 17, 0, 255), (1.0 / 127.0, 0, -128, 127), (0.25, -3, -8, 7)])
def test_affine_kernel_equals_aten_cpu_for_every_float_in_the_parity_domain(lib, scale, zp, qmin, qmax):
    """All float32 bit patterns with |x / scale| < 2^31 (everything else is excluded by the parity gate):
    the HIP kernel equals the ATen CPU operator the reference calls, bit for bit."""
    torch.set_num_threads(min(32, torch.get_num_threads()))
    chunk = 1 << 27
    s32 = float(np.float32(scale))
    limit = float(np.float32(2.0 ** 31) * np.float32(scale))
    y = torch.empty(chunk, dtype=torch.float32, device="cuda")
    for c in range(32):
        bits = torch.arange(c * chunk - (1 << 31), (c + 1) * chunk - (1 << 31), dtype=torch.int64, device="cuda")
        x = bits.to(torch.int32).view(torch.float32)
        del bits
        assert lib.mctq_fq_per_tensor_f32(x.data_ptr(), y.data_ptr(), chunk, s32, zp, qmin, qmax, _stream()) == 0
        x_cpu = x.cpu()
        want = torch.fake_quantize_per_tensor_affine(x_cpu, s32, zp, qmin, qmax)
        got = y.cpu()
        inside = x_cpu.abs() < limit                                    # NaN/inf compare False -> excluded
        same = (got.view(torch.int32) == want.view(torch.int32)) | ~inside
        if not bool(same.all()):
            i = int(torch.nonzero(~same)[0])
            raise AssertionError(f"x={x_cpu[i].item()!r} hip={got[i].item()!r} aten_cpu={want[i].item()!r}")
        # outside the domain the kernel saturates instead of inheriting the CPU path's int64-cast UB
        big = (x_cpu.abs() >= limit) & torch.isfinite(x_cpu)
        if bool(big.any()):
            sat = torch.where(x_cpu > 0, torch.tensor((qmax - zp) * s32), torch.tensor((qmin - zp) * s32))
            assert bool(((got == sat) | ~big).all())
        del x


def _table(lut, mult=128.0, cmin=-128.0, cmax=127.0):
    from mct_quantizers_amd.hip import native
    tab = native.build_lut_table(lut, mult, cmin, cmax)
    assert tab is not None
    return _dev(tab)


@pytest.mark.parametrize("lut_name", list(LUTS))
def test_decision_table_equals_literal_scan_for_every_float(lib, lut_name):
    """All 2^32 float32 inputs: the LDS decision-table kernel == the literal first-minimum scan kernel."""
    lut = LUTS[lut_name]
    lut_d, tab = _dev(np.asarray(lut, dtype=np.float32)), _table(lut)
    chunk = 1 << 28
    y_lit = torch.empty(chunk, dtype=torch.float32, device="cuda")
    y_tab = torch.empty(chunk, dtype=torch.float32, device="cuda")
    for c in range(16):
        bits = torch.arange(c * chunk - (1 << 31), (c + 1) * chunk - (1 << 31), dtype=torch.int64, device="cuda")
        x = bits.to(torch.int32).view(torch.float32)
        del bits
        # thr_div = 1, mult = 128: t = clamp(x * 128) sweeps every float in the clip range
        assert lib.mctq_lut_per_tensor_f32(x.data_ptr(), y_lit.data_ptr(), chunk, 1.0, 1.0, lut_d.data_ptr(),
                                           len(lut), 128.0, -128.0, 127.0, _stream()) == 0
        assert lib.mctq_lutt_per_tensor_f32(x.data_ptr(), y_tab.data_ptr(), chunk, 1.0, 1.0, tab.data_ptr(),
                                            tab.shape[0] - 1, 128.0, -128.0, 127.0, _stream()) == 0
        same = torch.equal(y_lit.view(torch.int32), y_tab.view(torch.int32))
        if not same:
            i = int(torch.nonzero(y_lit.view(torch.int32) != y_tab.view(torch.int32))[0])
            raise AssertionError(f"chunk {c}: x={x[i].item()!r} literal={y_lit[i].item()!r} table={y_tab[i].item()!r}")
        del x


def test_decision_table_unsigned_and_wide_codebooks(lib):
    from oracle import mctq_oracle as O
    rng = np.random.default_rng(3)
    cases = [([0.0, 13.0, 50.0, 90.0, 128.0, 200.0, 255.0, 256.0], False, 8),
             ([float(v) for v in range(-512, 512, 64)], True, 10),
             ([-8.0, -3.0, 2.0, 7.0], True, 4)]
    for lut, signed, B in cases:
        mult = float(2 ** (B - int(signed)))
        cmin, cmax = (float(-2 ** (B - 1)), float(2 ** (B - 1) - 1)) if signed else (0.0, float(2 ** B - 1))
        tab = _table(lut, mult, cmin, cmax)
        thr = 1.7
        x_np = (rng.standard_normal(200003).astype(np.float32) * np.float32(1.2))
        mid = (rng.integers(int(2 * cmin) - 4, int(2 * cmax) + 4, size=x_np.size).astype(np.float32) * np.float32(0.5)
               / np.float32(mult) * np.float32(thr))
        x_np = np.where(rng.integers(0, 3, size=x_np.size) == 0, mid, x_np).astype(np.float32)
        x = _dev(x_np)
        y = torch.empty_like(x)
        thr_div = float(np.float32(thr) + np.float32(1e-8))
        assert lib.mctq_lutt_per_tensor_f32(x.data_ptr(), y.data_ptr(), x.numel(), thr_div, float(np.float32(thr)),
                                            tab.data_ptr(), tab.shape[0] - 1, mult, cmin, cmax, _stream()) == 0
        want = O.lut_quantize(x_np, lut, np.asarray([thr], dtype=np.float32), signed, B, 1e-8)
        assert bits_equal(y.cpu().numpy(), want), first_mismatch(y.cpu().numpy(), want, x_np)


@pytest.mark.parametrize("lut_name", ["l3dup", "l16", "l256"])
@pytest.mark.parametrize("outer,C,inner", [(1, 3, 1), (50, 3, 1), (4, 6, 5), (2, 8, 100), (2, 6, 1024), (3, 5, 1028),
                                           (1, 16, 11008), (1, 3000, 3), (1, 2, 70000), (41, 64, 1), (3, 4096, 1)])
def test_abi_table_lut_per_channel_vs_oracle(lib, lut_name, outer, C, inner):
    from oracle import mctq_oracle as O
    rng = np.random.default_rng(C * 31 + inner + 1)
    lut = LUTS[lut_name]
    thr = rng.uniform(0.05, 4.0, size=C).astype(np.float32)
    shape = (outer, C, inner)
    x_np = _lut_inputs(rng, shape, thr.reshape(1, C, 1))
    x, t_d, tab = _dev(x_np), _dev(thr), _table(lut)
    y = torch.empty_like(x)
    rc = lib.mctq_lutt_per_channel_f32(x.data_ptr(), y.data_ptr(), outer, C, inner, t_d.data_ptr(), 1e-8,
                                       tab.data_ptr(), tab.shape[0] - 1, 128.0, -128.0, 127.0, _stream())
    assert rc == 0, lib.mctq_last_error()
    want = O.lut_quantize(x_np, lut, thr, True, 8, 1e-8, per_channel=True, channel_axis=1)
    assert bits_equal(y.cpu().numpy(), want), first_mismatch(y.cpu().numpy(), want, x_np)


@pytest.mark.parametrize("dt", ["float16", "bfloat16"])
@pytest.mark.parametrize("outer,C,inner", [(50, 3, 1), (4, 6, 5), (2, 8, 104), (2, 6, 2048), (3, 5, 1032), (1, 16, 11008),
                                           (41, 64, 1), (3, 4096, 1), (1, 3000, 3)])
def test_abi_table_lut_half_inputs_vs_oracle(lib, dt, outer, C, inner):
    """16-bit inputs to the LUT kernels: float32 arithmetic on the widened value, float32 output (the
    reference's weights-LUT chain promotes at the first division)."""
    from oracle import mctq_oracle as O
    code = {"float16": 1, "bfloat16": 2}[dt]
    rng = np.random.default_rng(C * 13 + inner)
    lut = LUTS["l16"]
    thr = rng.uniform(0.05, 4.0, size=C).astype(np.float32)
    shape = (outer, C, inner)
    xh = _dev(_lut_inputs(rng, shape, thr.reshape(1, C, 1))).to(getattr(torch, dt))
    x_np = xh.float().cpu().numpy()
    t_d, tab = _dev(thr), _table(lut)
    y = torch.empty(shape, dtype=torch.float32, device="cuda")
    rc = lib.mctq_lutt_per_channel(xh.data_ptr(), y.data_ptr(), outer, C, inner, code, t_d.data_ptr(), 1e-8,
                                   tab.data_ptr(), tab.shape[0] - 1, 128.0, -128.0, 127.0, _stream())
    assert rc == 0, lib.mctq_last_error()
    want = O.lut_quantize(x_np, lut, thr, True, 8, 1e-8, per_channel=True, channel_axis=1)
    assert finite_equal(y.cpu().numpy(), want, x_np), first_mismatch(y.cpu().numpy(), want, x_np)
    # per-tensor, with the per-op half roundings of an activation tensor (step_round)
    y1 = torch.empty(xh.numel(), dtype=torch.float32, device="cuda")
    thr_div = float(torch.tensor([2.0 + 1e-8], dtype=torch.float64).to(getattr(torch, dt)).item())
    rc = lib.mctq_lutt_per_tensor(xh.data_ptr(), y1.data_ptr(), xh.numel(), code, code, thr_div, 2.0, tab.data_ptr(),
                                  tab.shape[0] - 1, 128.0, -128.0, 127.0, _stream())
    assert rc == 0, lib.mctq_last_error()
    want1 = O.lut_quantize(x_np.reshape(-1), lut, 2.0, True, 8, 1e-8, step_dtype=dt)
    assert finite_equal(y1.cpu().numpy(), want1, x_np.reshape(-1)), first_mismatch(y1.cpu().numpy(), want1, x_np.reshape(-1))


def test_fast_division_is_exact(lib):
    """Every float32 numerator x 48 divisors: the shared-divisor division equals IEEE '/' bit for bit."""
    rng = np.random.default_rng(99)
    special = [1.0, 2.0, 0.5, 3.0, 1.0 / 3.0, 0.1, 1e-8, 1.00000001e-8, 127.0, 1.9999999, 1.0000001, 0.99999994,
               3.4028235e10, 2.0 ** -59, 2.0 ** 59, 1e-20, 7.0, 1.5, 4.7683716e-07 + 1e-8, 0.33, 11.3, 5e-5]
    rand = np.exp(rng.uniform(np.log(1e-6), np.log(1e6), size=48 - len(special)))
    div = np.asarray(special + list(rand), dtype=np.float32)
    d = _dev(div)
    bad = torch.zeros(div.size, dtype=torch.int64, device="cuda")
    rc = lib.mctq_selftest_division(d.data_ptr(), div.size, bad.data_ptr(), _stream())
    assert rc == 0, lib.mctq_last_error()
    torch.cuda.synchronize()
    bad = bad.cpu().numpy()
    assert not bad.any(), {float(div[i]): int(bad[i]) for i in np.flatnonzero(bad)}


@pytest.mark.parametrize("dt", ["float32", "float16", "bfloat16"])
@pytest.mark.parametrize("outer,C,inner", [(7, 3, 1), (33, 5, 2), (2, 8, 100), (2, 6, 1024), (1, 64, 4096), (1, 5000, 7),
                                           (19, 256, 1)])
def test_integer_codes_equal_the_oracle_index(lib, dt, outer, C, inner):
    """int8 / uint8 code outputs == the oracle's clamp index, for per-channel and per-tensor, all storage types."""
    from oracle import mctq_oracle as O
    rng = np.random.default_rng(C + inner)
    tdt = getattr(torch, dt)
    code_in = {"float32": 0, "float16": 1, "bfloat16": 2}[dt]
    for qmin, qmax, tcode, code_dt in ((-128, 127, torch.int8, 0), (0, 255, torch.uint8, 1), (-8, 7, torch.int8, 0)):
        scales = rng.uniform(0.01, 0.2, size=C).astype(np.float32)
        zps = rng.integers(qmin, qmax + 1, size=C).astype(np.int32)
        shape = (outer, C, inner)
        x32 = _tie_heavy(rng, shape, scales.reshape(1, C, 1), zps.reshape(1, C, 1).astype(np.float32), qmin, qmax)
        x = _dev(x32).to(tdt)
        x_np = x.float().cpu().numpy()
        codes = torch.empty(shape, dtype=tcode, device="cuda")
        s_d, z_d = _dev(scales), _dev(zps)
        rc = lib.mctq_fq_codes_per_channel(x.data_ptr(), codes.data_ptr(), outer, C, inner, code_in, code_dt,
                                           s_d.data_ptr(), z_d.data_ptr(), qmin, qmax, _stream())
        assert rc == 0, lib.mctq_last_error()
        _, q_want = O.fake_quant_affine(x_np, scales, zps, qmin, qmax, axis=1, return_index=True)
        finite = np.isfinite(x_np)
        assert np.array_equal(codes.cpu().numpy().astype(np.int64)[finite], q_want[finite])
        rc = lib.mctq_fq_codes_per_tensor(x.data_ptr(), codes.data_ptr(), x.numel(), code_in, code_dt, float(scales[0]),
                                          int(zps[0]), qmin, qmax, _stream())
        assert rc == 0, lib.mctq_last_error()
        _, q_want = O.fake_quant_affine(x_np, scales[0], zps[0], qmin, qmax, return_index=True)
        assert np.array_equal(codes.cpu().numpy().astype(np.int64)[finite], q_want[finite])
    assert lib.mctq_fq_codes_per_tensor(x.data_ptr(), codes.data_ptr(), 4, 0, 0, 1.0, 0, -200, 127, None) == -10001


def test_quantize_to_codes_dequantizes_bit_exactly(lib):
    import mct_quantizers_amd as mq
    x = torch.randn(4096, 512, device="cuda") * 2
    q = mq.pytorch_quantizers.WeightsSymmetricInferableQuantizer(8, [0.5 + 0.001 * i for i in range(4096)], True, 0)
    codes, s, z = q.quantize_to_codes(x)
    assert codes.dtype == torch.int8 and codes.shape == x.shape
    assert torch.equal((codes.float() - z.float().reshape(-1, 1)) * s.reshape(-1, 1), q(x))
    qa = mq.pytorch_quantizers.ActivationUniformInferableQuantizer(8, [-2.5], [3.1])
    codes, s, z = qa.quantize_to_codes(x)
    assert codes.dtype == torch.uint8
    assert torch.equal((codes.float() - z) * torch.tensor(s, dtype=torch.float32, device="cuda"), qa(x))


def test_abi_argument_errors(lib):
    x = torch.zeros(16, device="cuda")
    y = torch.zeros(16, device="cuda")
    assert lib.mctq_fq_per_tensor_f32(x.data_ptr(), y.data_ptr(), -1, 1.0, 0, 0, 255, None) == -10001
    assert b"n < 0" in lib.mctq_last_error()
    assert lib.mctq_fq_per_tensor_f32(x.data_ptr(), y.data_ptr(), 16, 1.0, 0, 5, 4, None) == -10001
    assert lib.mctq_fq_per_tensor_f32(None, y.data_ptr(), 16, 1.0, 0, 0, 255, None) == -10001
    assert lib.mctq_lut_per_tensor_f32(x.data_ptr(), y.data_ptr(), 16, 1.0, 1.0, None, 4, 128.0, -128.0, 127.0,
                                       None) == -10001
    assert lib.mctq_lut_per_tensor_f32(x.data_ptr(), y.data_ptr(), 16, 1.0, 1.0, x.data_ptr(), 4, 100.0, -128.0,
                                       127.0, None) == -10001
    assert lib.mctq_set_tuning(b"bogus", 1) == -10001


# ---------------------------------------------------------------------------------------------
# 3. the five BASELINE configurations at full size: SHA-256 equals the REFERENCE's output digest
# ---------------------------------------------------------------------------------------------

@pytest.mark.parametrize("cfg", ["cfg1", "cfg2", "cfg3", "cfg4", "cfg5"])
def test_full_size_config_matches_reference_digest(lib, cfg):
    import mct_quantizers_amd as mq
    from mct_quantizers_amd import workloads
    rec = load_json("full_sha.json")["configs"][cfg]
    x_np = workloads.make_input(cfg, batch=8)
    assert list(x_np.shape) == rec["shape"]
    assert hashlib.sha256(x_np.tobytes()).hexdigest() == rec["x_sha256"]
    wl = workloads.make_workload(cfg, x_np)
    q = getattr(mq.pytorch_quantizers, wl.quantizer)(**wl.kwargs)
    x = _dev(x_np)
    y = q(x)
    y_np = y.cpu().numpy()
    assert hashlib.sha256(np.ascontiguousarray(y_np).tobytes()).hexdigest() == rec["y_sha256"]
    # size-independent properties
    if "LUT" not in wl.quantizer:
        y2 = q(y.clone())
        assert torch.equal(y2, y), "fake-quant must be idempotent"
        row = y_np.reshape(y_np.shape[0], -1)[0]
        assert np.unique(row).size <= 2 ** wl.kwargs["num_bits"]
    else:
        row = y_np[0]
        thr0 = np.float32(wl.kwargs["threshold"][0])
        allowed = np.asarray(wl.kwargs["lut_values"], dtype=np.float32) / np.float32(128.0) * thr0
        assert np.all(np.isin(row, allowed))
        assert torch.equal(q(y.clone()), y), "codebook values quantize to themselves"
    # sortedness: the quantizer is monotone -- a sorted row stays sorted (parameters of row 0 / the whole tensor)
    flat = x.reshape(x.shape[0], -1) if x.dim() > 1 else x.reshape(1, -1)
    srt = torch.sort(flat, dim=1).values.reshape(x.shape)
    ys = q(srt).reshape(flat.shape)
    assert bool((ys[:, 1:] >= ys[:, :-1]).all()), "monotone in the input"


# ---------------------------------------------------------------------------------------------
# 4. behaviour around the kernels: tuning variants, streams, graphs, side effects, errors
# ---------------------------------------------------------------------------------------------

def test_tuning_variants_do_not_change_results(lib):
    from mct_quantizers_amd.hip import native
    from oracle import mctq_oracle as O
    rng = np.random.default_rng(11)
    C, inner = 37, 2052
    scales = rng.uniform(0.01, 0.5, size=C).astype(np.float32)
    zps = np.zeros(C, dtype=np.int32)
    x_np = _tie_heavy(rng, (C, inner), scales.reshape(C, 1), np.float32(0), -128, 127)
    want = O.fake_quant_affine(x_np, scales, zps, -128, 127, axis=0)
    x, s_d, z_d = _dev(x_np), _dev(scales), _dev(zps)
    try:
        for key, bad in (("nt", 0), ("nt", 3), ("unroll", 8), ("unroll", 3), ("heavy_unroll", 8), ("heavy_persistent", 1)):
            with pytest.raises(RuntimeError):                  # values whose kernels are not built (ABI v8) are refused
                native.set_tuning(key, bad)
        for nt in (1, 2):
            for unroll in (1, 2, 4):
                native.set_tuning("nt", nt)
                native.set_tuning("unroll", unroll)
                y = torch.empty_like(x)
                rc = lib.mctq_fq_per_channel_f32(x.data_ptr(), y.data_ptr(), 1, C, inner, s_d.data_ptr(),
                                                 z_d.data_ptr(), -128, 127, _stream())
                assert rc == 0
                assert bits_equal(y.cpu().numpy(), want), (nt, unroll)
                y = torch.empty_like(x)
                rc = lib.mctq_fq_per_tensor_f32(x.data_ptr(), y.data_ptr(), x.numel(), float(scales[0]), 3, 0, 255,
                                                _stream())
                assert rc == 0
                assert bits_equal(y.cpu().numpy(), O.fake_quant_affine(x_np, scales[0], 3, 0, 255)), (nt, unroll)
    finally:
        native.set_tuning("nt", 1)
        native.set_tuning("unroll", 4)


def test_lut_tuning_variants_do_not_change_results(lib):
    from mct_quantizers_amd.hip import native
    from oracle import mctq_oracle as O
    rng = np.random.default_rng(12)
    C, inner = 9, 11008
    lut = LUTS["l16"]
    thr = rng.uniform(0.05, 4.0, size=C).astype(np.float32)
    x_np = _lut_inputs(rng, (1, C, inner), thr.reshape(1, C, 1))
    want = O.lut_quantize(x_np, lut, thr, True, 8, 1e-8, per_channel=True, channel_axis=1)
    x, t_d, tab, lut_d = _dev(x_np), _dev(thr), _table(lut), _dev(np.asarray(lut, dtype=np.float32))
    try:
        for nt, hu, pers in [(n_, h_, 0) for n_ in (1, 2) for h_ in (1, 2, 4, 0)]:
            if True:
                native.set_tuning("nt", nt)
                native.set_tuning("heavy_unroll", hu)
                for use_table in (True, False):
                    y = torch.empty_like(x)
                    if use_table:
                        rc = lib.mctq_lutt_per_channel_f32(x.data_ptr(), y.data_ptr(), 1, C, inner, t_d.data_ptr(), 1e-8,
                                                           tab.data_ptr(), tab.shape[0] - 1, 128.0, -128.0, 127.0,
                                                           _stream())
                    else:
                        rc = lib.mctq_lut_per_channel_f32(x.data_ptr(), y.data_ptr(), 1, C, inner, t_d.data_ptr(), 1e-8,
                                                          lut_d.data_ptr(), len(lut), 128.0, -128.0, 127.0, _stream())
                    assert rc == 0, lib.mctq_last_error()
                    assert bits_equal(y.cpu().numpy(), want), (nt, hu, pers, use_table)
                    # per-tensor launch shapes of the same ops
                    y = torch.empty_like(x)
                    if use_table:
                        rc = lib.mctq_lutt_per_tensor_f32(x.data_ptr(), y.data_ptr(), x.numel(), float(thr[0] + np.float32(1e-8)),
                                                          float(thr[0]), tab.data_ptr(), tab.shape[0] - 1, 128.0, -128.0,
                                                          127.0, _stream())
                    else:
                        rc = lib.mctq_lut_per_tensor_f32(x.data_ptr(), y.data_ptr(), x.numel(), float(thr[0] + np.float32(1e-8)),
                                                         float(thr[0]), lut_d.data_ptr(), len(lut), 128.0, -128.0, 127.0,
                                                         _stream())
                    assert rc == 0, lib.mctq_last_error()
                    assert bits_equal(y.cpu().numpy(), O.lut_quantize(x_np, lut, thr[:1], True, 8, 1e-8)), \
                        (nt, hu, pers, use_table, "per-tensor")
    finally:
        native.set_tuning("nt", 1)
        native.set_tuning("heavy_unroll", 0)


def test_one_vector_window_tiles_where_the_parameter_window_of_four_would_not_fit_lds(lib):
    """The dispatcher's last branch (csrc/mctq_kernels.hpp: launch_channels): per-channel along the fastest axis with so many
    channels that the parameter window of a 4-lane-vector tile exceeds 64 KiB of LDS -> tiles of ONE lane-vector.  The
    launch log of the whole suite never reached it before this test (profiles/r05/launch_variants_all.log)."""
    from mct_quantizers_amd.hip import native
    from oracle import mctq_oracle as O
    rng = np.random.default_rng(77)
    # (a) affine, bfloat16 storage: 8192-element tiles touch 8192 rows of one element, 8193 x 12 B > 64 KiB
    rows, C = 700, 9001                                      # C % 8 != 0: not the lastaxis shape
    scales = rng.uniform(0.01, 0.5, size=C).astype(np.float32)
    zps = rng.integers(0, 16, size=C).astype(np.int32)
    x32 = _tie_heavy(rng, (rows, C), scales.reshape(1, C), zps.reshape(1, C).astype(np.float32), 0, 15)
    xb = _dev(x32).to(torch.bfloat16)
    yb = torch.empty_like(xb)
    s_d, z_d = _dev(scales), _dev(zps)
    rc = lib.mctq_fq_per_channel(xb.data_ptr(), yb.data_ptr(), rows, C, 1, native.DT_BF16, s_d.data_ptr(), z_d.data_ptr(), 0, 15,
                                 _stream())
    assert rc == 0, lib.mctq_last_error()
    assert native.last_launch() == "window_kernel<vector><AffineOp,in2B,out2B,U=1,NT=2>", native.last_launch()
    xw = xb.float().cpu().numpy()
    want = torch.from_numpy(O.fake_quant_affine(xw, scales, zps, 0, 15, axis=1)).to(torch.bfloat16)
    assert torch.equal(yb.cpu(), want)
    # (b) decision-table LUT, float32: 4097 rows x 16 B + the 4 KB table > 64 KiB
    rows, C = 500, 4099
    lut = LUTS["l16"]
    thr = rng.uniform(0.05, 4.0, size=C).astype(np.float32)
    x_np = _lut_inputs(rng, (rows, C), thr.reshape(1, C))
    x, t_d, tab = _dev(x_np), _dev(thr), _table(lut)
    y = torch.empty_like(x)
    rc = lib.mctq_lutt_per_channel_f32(x.data_ptr(), y.data_ptr(), rows, C, 1, t_d.data_ptr(), 1e-8, tab.data_ptr(),
                                       tab.shape[0] - 1, 128.0, -128.0, 127.0, _stream())
    assert rc == 0, lib.mctq_last_error()
    assert native.last_launch().startswith("window_kernel<vector><LutTableOp,in4B,out4B,U=1,"), native.last_launch()
    want = O.lut_quantize(x_np, lut, thr, True, 8, 1e-8, per_channel=True, channel_axis=1)
    assert bits_equal(y.cpu().numpy(), want), first_mismatch(y.cpu().numpy(), want, x_np)
    # (c) literal scan (single-variant op), same shape: registers-codebook kernel through the same branch
    lut_d = _dev(np.asarray(lut, dtype=np.float32))
    y2 = torch.empty_like(x)
    rc = lib.mctq_lut_per_channel_f32(x.data_ptr(), y2.data_ptr(), rows, C, 1, t_d.data_ptr(), 1e-8, lut_d.data_ptr(), len(lut),
                                      128.0, -128.0, 127.0, _stream())
    assert rc == 0, lib.mctq_last_error()
    assert native.last_launch() == "window_kernel<vector><LutOp<registers>,in4B,out4B,U=1,NT=1>", native.last_launch()
    assert bits_equal(y2.cpu().numpy(), want)


def test_cached_store_threshold_does_not_change_results(lib):
    import mct_quantizers_amd as mq
    from mct_quantizers_amd.hip import native
    Q = mq.pytorch_quantizers
    x = torch.randn(300, 2048, device="cuda")
    xc = x.contiguous(memory_format=torch.contiguous_format)
    qs = [Q.WeightsSymmetricInferableQuantizer(8, [1.0 + 0.01 * i for i in range(300)], True, 0),
          Q.WeightsSymmetricInferableQuantizer(8, [1.0 + 0.001 * i for i in range(2048)], True, 1),
          Q.ActivationUniformInferableQuantizer(8, [-2.5], [3.1]),
          Q.WeightsLUTSymmetricInferableQuantizer(4, LUTS["l16"], [4.0] * 300, True, 0, 2)]
    base = [q(xc.clone()) for q in qs]
    try:
        native.set_tuning("cached_store_max_mb", 64)
        for q, b in zip(qs, base):
            assert torch.equal(q(xc.clone()), b)
    finally:
        native.set_tuning("cached_store_max_mb", 32)
    # bfloat16 storage under both store policies, against the oracle (not against itself)
    from oracle import oracle_call
    xb = xc.clone().bfloat16()
    kw = dict(num_bits=8, threshold=[1.0 + 0.01 * i for i in range(300)], per_channel=True, channel_axis=0)
    want = oracle_call("WeightsSymmetricInferableQuantizer", kw, xb.float().cpu().numpy(), in_dtype="bfloat16")
    for mb in (64, 0, 32):
        native.set_tuning("cached_store_max_mb", mb)
        try:
            got = Q.WeightsSymmetricInferableQuantizer(**kw)(xb.clone())
            assert got.dtype == torch.bfloat16 and bits_equal(got.float().cpu().numpy(), want), mb
        finally:
            native.set_tuning("cached_store_max_mb", 32)


def test_side_stream_and_graph_capture(lib):
    import mct_quantizers_amd as mq
    q = mq.pytorch_quantizers.ActivationUniformInferableQuantizer(8, [-2.5], [3.1])
    x = torch.randn(8, 3, 64, 64, device="cuda")
    ref = q(x)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        y = q(x)
    side.synchronize()
    assert torch.equal(y, ref)
    # hipGraph: capture once, replay on new data
    static_x = x.clone()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        static_y = q(static_x)
    x2 = torch.randn_like(x)
    static_x.copy_(x2)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(static_y, q(x2))


def test_graph_capture_of_a_wrapped_layer_stack(lib):
    """A small quantized model (per-channel affine + LUT weights, activation holders) captured into one
    hipGraph replays to the same outputs as eager execution."""
    import mct_quantizers_amd as mq
    Q = mq.pytorch_quantizers
    torch.manual_seed(3)
    lin1, lin2 = torch.nn.Linear(256, 512).cuda(), torch.nn.Linear(512, 64).cuda()
    thr1 = [float(v) for v in lin1.weight.detach().abs().amax(dim=1)]
    thr2 = [float(v) for v in lin2.weight.detach().abs().amax(dim=1)]
    lut = [-128.0, -96.0, -64.0, -40.0, -24.0, -12.0, -5.0, 0.0, 5.0, 12.0, 24.0, 40.0, 64.0, 96.0, 120.0, 127.0]
    model = torch.nn.Sequential(
        mq.PytorchQuantizationWrapper(lin1, {"weight": Q.WeightsSymmetricInferableQuantizer(8, thr1, True, 0)}),
        mq.PytorchActivationQuantizationHolder(Q.ActivationUniformInferableQuantizer(8, [-3.0], [3.0])),
        mq.PytorchQuantizationWrapper(lin2, {"weight": Q.WeightsLUTSymmetricInferableQuantizer(4, lut, thr2, True, 0, 2)}),
        mq.PytorchActivationQuantizationHolder(Q.ActivationLutPOTInferableQuantizer(4, lut, [4.0], True)))
    x = torch.randn(32, 256, device="cuda")
    with torch.no_grad():
        want = model(x)
        static_x = x.clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            model(static_x)                                   # warm-up on the capture stream
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            static_y = model(static_x)
        x2 = torch.randn_like(x)
        static_x.copy_(x2)
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(static_y, model(x2))
        static_x.copy_(x)
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(static_y, want)


def test_weights_quantizer_side_effects_and_reuse(lib):
    import mct_quantizers_amd as mq
    q = mq.pytorch_quantizers.WeightsSymmetricInferableQuantizer(8, [1.0, 2.0, 0.5], True, 0)
    w = torch.nn.Parameter(torch.randn(3, 50, device="cuda"))
    assert w.requires_grad
    y = q(w)
    assert not w.requires_grad and not y.requires_grad        # reference flips requires_grad on the input
    q.enable_reuse_quantizer()
    a = q(w)
    b = q(torch.zeros_like(w))
    assert b is a and q.resue_outputs is a and not q.quantizer_first_run
    q.disable_reuse_quantizer()
    assert q(torch.zeros_like(w)).abs().sum().item() == 0.0


def test_activation_quantizer_runs_without_grad(lib):
    import mct_quantizers_amd as mq
    q = mq.pytorch_quantizers.ActivationSymmetricInferableQuantizer(8, [4.0], True)
    x = torch.randn(4, 16, device="cuda", requires_grad=True)
    y = q(x)
    assert not y.requires_grad and x.requires_grad


def test_non_dense_and_permuted_inputs(lib):
    import mct_quantizers_amd as mq
    from oracle import oracle_call
    kw = dict(num_bits=4, threshold=[0.5, 1.0, 2.0, 4.0], per_channel=True, channel_axis=1)
    q = mq.pytorch_quantizers.WeightsPOTInferableQuantizer(**kw)
    base = torch.randn(6, 4, 10, 12, device="cuda")
    want = oracle_call("WeightsPOTInferableQuantizer", kw, base.cpu().numpy())
    for x in (base, base.contiguous(memory_format=torch.channels_last), base.permute(0, 1, 3, 2).contiguous().permute(0, 1, 3, 2)):
        y = q(x.clone(memory_format=torch.preserve_format))
        assert y.stride() == x.stride()
        assert bits_equal(y.cpu().numpy(), want)
    sliced = base[:, :, ::2, :]                                # gaps: not dense -> compacted
    want_s = oracle_call("WeightsPOTInferableQuantizer", kw, sliced.cpu().numpy())
    assert bits_equal(q(sliced).cpu().numpy(), want_s)


def test_fuzz_shapes_axes_layouts_and_dtypes_against_aten_cpu(lib):
    """Seeded fuzz: random ranks, shapes, channel axes, dimension permutations, storage types and slices;
    the HIP result must equal ATen's CPU operator on the same (finite, in-domain) tensor, strides included."""
    import mct_quantizers_amd as mq
    Q = mq.pytorch_quantizers
    import os
    rng = np.random.default_rng(int(os.environ.get("MCTQ_FUZZ_SEED", "2024")))       # soak runs: other seeds, more cases
    for case in range(int(os.environ.get("MCTQ_FUZZ_CASES", "160"))):
        rank = int(rng.integers(1, 6))
        shape = [int(rng.choice([1, 2, 3, 4, 5, 7, 8, 16, 33, 64])) for _ in range(rank)]
        if rng.random() < 0.25:
            shape[int(rng.integers(0, rank))] = int(rng.choice([257, 1024, 1030, 4096]))
        axis = int(rng.integers(0, rank))
        dt = [torch.float32, torch.float32, torch.float16, torch.bfloat16][int(rng.integers(0, 4))]
        C = shape[axis]
        bits = int(rng.choice([2, 4, 8]))
        if int(np.prod(shape, dtype=np.int64)) > (1 << 26):      # bound every case (256 MiB), whatever the seed draws
            continue
        x = (torch.from_numpy(rng.standard_normal(shape).astype(np.float32)) * 3).to(dt)
        perm = list(rng.permutation(rank))
        x = x.permute(perm).contiguous().permute(list(np.argsort(perm)))      # same logical shape, permuted storage
        if rng.random() < 0.2 and shape[0] > 1:
            x = x[::2]                                                            # gaps: not dense
            C = x.shape[axis]
        kind = int(rng.integers(0, 3))
        if kind == 0:
            q = Q.WeightsSymmetricInferableQuantizer(bits, [float(v) for v in rng.uniform(0.2, 6.0, size=C)], True, axis)
            ref = lambda t: torch.fake_quantize_per_channel_affine(t, q.scales.cpu(), q.zero_points.cpu(), axis,
                                                                   q.min_quantized_domain, q.max_quantized_domain)
        elif kind == 1:
            lo = [float(v) for v in rng.uniform(-4.0, -0.1, size=C)]
            hi = [float(v) for v in rng.uniform(0.1, 5.0, size=C)]
            q = Q.WeightsUniformInferableQuantizer(bits, lo, hi, True, axis)
            ref = lambda t: torch.fake_quantize_per_channel_affine(t, q.scales.cpu(), q.zero_points.cpu(), axis,
                                                                   0, 2 ** bits - 1)
        else:
            q = Q.ActivationUniformInferableQuantizer(bits, [-2.5], [3.1])
            ref = lambda t: torch.fake_quantize_per_tensor_affine(t, q.scale, q.zero_point, 0, 2 ** bits - 1)
        want = ref(x.clone())
        got = q(x.cuda())
        info = (case, tuple(x.shape), x.stride(), axis, dt, kind)
        assert got.dtype == want.dtype and got.shape == want.shape, info
        assert got.stride() == want.stride(), info             # ATen's preserve-format strides, gapped views included
        assert torch.equal(got.cpu().float().view(torch.int32), want.float().view(torch.int32)), info
        if x.is_contiguous() or kind != 2:
            pass
        # dense inputs keep their strides (ATen semantics); gapped ones are compacted
        xc = x.cuda()
        from mct_quantizers_amd.hip.ops import _is_dense
        if _is_dense(xc):
            assert got.stride() == xc.stride(), info


def test_tensors_beyond_2_pow_31_elements(lib):
    """64-bit indexing: more than 2^31 elements per tensor (8 GiB each), per-tensor, per-channel rows, per-channel
    window (ragged inner) and channel-last shapes, against ATen's HIP operator (itself checked elsewhere)."""
    import mct_quantizers_amd as mq
    Q = mq.pytorch_quantizers
    free, _ = torch.cuda.mem_get_info()
    if free < 60 * (1 << 30):
        pytest.skip("needs ~40 GiB of free HBM")
    n = (1 << 31) + (1 << 22) + 3
    x = torch.empty(n, dtype=torch.float32, device="cuda")
    x.uniform_(-3.0, 3.0)
    qa = Q.ActivationUniformInferableQuantizer(8, [-2.5], [3.1])
    y = qa(x)
    want = torch.fake_quantize_per_tensor_affine(x, qa.scale, qa.zero_point, 0, 255)
    assert torch.equal(y, want)
    del y, want
    # rows: [C=2051][inner=1048576 + ...]: take a prefix that factors
    C, inner = 2051, 1048576
    xv = x[: C * inner].view(C, inner)
    thr = [1.0 + 0.001 * (i % 500) for i in range(C)]
    qw = Q.WeightsSymmetricInferableQuantizer(8, thr, True, 0)
    y = qw(xv)
    want = torch.fake_quantize_per_channel_affine(xv, qw.scales, qw.zero_points, 0, -128, 127)
    assert torch.equal(y, want)
    del y, want
    # window shape beyond 2^32 bytes: inner = 1021 (ragged), C = 2103443 rows
    inner = 1021
    C = (1 << 31) // inner + 3
    xv = x[: C * inner].view(C, inner)
    qw = Q.WeightsSymmetricInferableQuantizer(8, [1.0 + 0.001 * (i % 500) for i in range(C)], True, 0)
    y = qw(xv)
    want = torch.fake_quantize_per_channel_affine(xv, qw.scales, qw.zero_points, 0, -128, 127)
    assert torch.equal(y, want)
    del y, want
    # channel-last beyond 2^31 elements
    C = 4096
    rows = n // C
    xv = x[: rows * C].view(rows, C)
    qw = Q.WeightsSymmetricInferableQuantizer(8, [1.0 + 0.001 * (i % 500) for i in range(C)], True, 1)
    y = qw(xv)
    want = torch.fake_quantize_per_channel_affine(xv, qw.scales, qw.zero_points, 1, -128, 127)
    assert torch.equal(y, want)
    del y, want, x, xv

    # beyond 2^32 elements (16 GiB per tensor): the 64-bit index variants of the window / lastaxis kernels
    free, _ = torch.cuda.mem_get_info()
    if free < 120 * (1 << 30):
        print("second phase (> 2^32 elements) skipped: free HBM", free >> 30, "GiB")
        return
    print("running the > 2^32-element phase")
    n = (1 << 32) + (1 << 22) + 5
    x = torch.empty(n, dtype=torch.float32, device="cuda")
    x.uniform_(-3.0, 3.0)
    y = qa(x)
    want = torch.fake_quantize_per_tensor_affine(x, qa.scale, qa.zero_point, 0, 255)
    assert torch.equal(y, want)
    del y, want
    inner = 1021
    C = (1 << 32) // inner + 3
    xv = x[: C * inner].view(C, inner)
    qw = Q.WeightsSymmetricInferableQuantizer(8, [1.0 + 0.001 * (i % 500) for i in range(C)], True, 0)
    y = qw(xv)
    want = torch.fake_quantize_per_channel_affine(xv, qw.scales, qw.zero_points, 0, -128, 127)
    assert torch.equal(y, want)
    del y, want
    C = 4096
    rows = n // C
    xv = x[: rows * C].view(rows, C)
    qw = Q.WeightsSymmetricInferableQuantizer(8, [1.0 + 0.001 * (i % 500) for i in range(C)], True, 1)
    y = qw(xv)
    want = torch.fake_quantize_per_channel_affine(xv, qw.scales, qw.zero_points, 1, -128, 127)
    assert torch.equal(y, want)


def test_loud_failures(lib, monkeypatch):
    import mct_quantizers_amd as mq
    from mct_quantizers_amd.hip import native, ops
    q = mq.pytorch_quantizers.ActivationSymmetricInferableQuantizer(8, [4.0], True)
    with pytest.raises(NotImplementedError):
        q(torch.zeros(8, device="cuda", dtype=torch.int32))                # not a kernel type (float64 is, since round 2)
    qc = mq.pytorch_quantizers.WeightsSymmetricInferableQuantizer(8, [1.0, 2.0], True, 0)
    with pytest.raises(RuntimeError):
        qc(torch.zeros(3, 4, device="cuda"))                   # 2 scales for 3 channels
    # a missing library is an error, never a silent fallback -- whichever binding is in use: the compiled one
    # links libmctq_hip.so, so without the library it cannot load either (a warning says so) and the ctypes
    # route then raises
    monkeypatch.setattr(native, "_lib", None)
    monkeypatch.setattr(native, "_fast", None)
    monkeypatch.setattr(native, "_fast_tried", False)
    monkeypatch.setattr(ops, "_FAST", None)
    monkeypatch.setattr(ops, "_FAST_READY", False)
    monkeypatch.setenv("MCTQ_HIP_LIB", "/nonexistent/libmctq_hip.so")
    import os
    import warnings
    if os.environ.get("MCTQ_BINDING") == "ctypes" or native.TRACE:
        q2 = mq.pytorch_quantizers.ActivationSymmetricInferableQuantizer(8, [4.0], True)
    else:
        with pytest.warns(UserWarning, match="compiled binding not loaded"):
            q2 = mq.pytorch_quantizers.ActivationSymmetricInferableQuantizer(8, [4.0], True)
    with pytest.raises(native.NativeLibraryError):
        q2(torch.zeros(8, device="cuda"))
    if not native.TRACE:                               # (MCTQ_ROCTX=1 routes everything through ctypes by design)
        monkeypatch.setenv("MCTQ_BINDING", "compiled")
        monkeypatch.setattr(native, "_fast_tried", False)
        with pytest.raises(native.NativeLibraryError):
            native.fast()


# ---------------------------------------------------------------------------------------------
# 4-bit packed codes (MCTQ_CODE_I4 / MCTQ_CODE_U4)
# ---------------------------------------------------------------------------------------------

@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize("shape,axis", [((64, 4096), 0), ((3, 5, 64), 1), ((7, 40), 1), ((33, 1024), 1), ((5, 8, 16), 0),
                                        ((4096,), None), ((3, 7, 8), None)])
def test_packed_4bit_codes_match_the_oracle_index(lib, dtype, shape, axis):
    import mct_quantizers_amd as mq
    from mct_quantizers_amd.hip import ops
    from oracle import oracle_call, mctq_oracle as O
    Q = mq.pytorch_quantizers
    rng = np.random.default_rng(int(np.prod(shape)) + (axis or 0))
    x32 = torch.from_numpy((rng.standard_normal(shape) * 1.5).astype(np.float32)).to(dtype)
    x_np = x32.float().numpy()
    if axis is None:
        for cls, kw in ((Q.ActivationPOTInferableQuantizer, dict(num_bits=4, threshold=[2.0], signed=True)),
                        (Q.ActivationUniformInferableQuantizer, dict(num_bits=4, min_range=[-1.0], max_range=[2.0])),
                        (Q.ActivationSymmetricInferableQuantizer, dict(num_bits=3, threshold=[1.7], signed=False))):
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                q = cls(**kw)
            packed, scale, zp = q.quantize_to_codes(x32.cuda(), packed4=True)
            _, idx = oracle_call(cls.__name__, kw, x_np, return_index=True)
            assert packed.dtype == torch.uint8 and packed.numel() == x32.numel() // 2
            assert np.array_equal(packed.cpu().numpy().reshape(-1), O.pack4(idx)), cls.__name__
            # and the CPU route packs identically
            assert torch.equal(q.quantize_to_codes(x32, packed4=True)[0].reshape(-1), packed.cpu().reshape(-1))
            # unpacked codes dequantize to the fake-quantized tensor (computed in float32, as the codes contract says)
            codes = ops.unpack4(packed.cpu(), q.min_quantized_domain < 0, shape)
            deq = (codes.float() - zp) * torch.tensor(scale, dtype=torch.float64).to(torch.float32)
            assert torch.equal(deq.to(dtype), q(x32.cuda()).cpu())
    else:
        C = shape[axis]
        thr = [float(v) for v in rng.uniform(0.5, 3.0, C)]
        kw = dict(num_bits=4, threshold=thr, per_channel=True, channel_axis=axis)
        q = Q.WeightsSymmetricInferableQuantizer(**kw)
        packed, scales, zps = q.quantize_to_codes(x32.cuda(), packed4=True)
        _, idx = oracle_call("WeightsSymmetricInferableQuantizer", kw, x_np, return_index=True)
        assert np.array_equal(packed.cpu().numpy().reshape(-1), O.pack4(idx))
        assert torch.equal(q.quantize_to_codes(x32, packed4=True)[0].reshape(-1), packed.cpu().reshape(-1))
        if x32.shape[-1] % 2 == 0:
            assert packed.shape == tuple(shape[:-1]) + (shape[-1] // 2,)


def test_packed_4bit_codes_reject_unsupported_layouts_and_domains(lib):
    from mct_quantizers_amd.hip import native, ops
    x = torch.randn(6, 12, device="cuda")
    s = torch.ones(6, device="cuda")
    z = torch.zeros(6, dtype=torch.int32, device="cuda")
    with pytest.raises(ValueError):
        ops.fq_codes(x, s, z, 0, -128, 127, packed4=True)                  # 8-bit domain
    with pytest.raises(RuntimeError, match="inner % 8"):
        ops.fq_codes(x, s, z, 0, -8, 7, packed4=True)                      # inner = 12: not a multiple of 8
    y = torch.empty(36, dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    assert lib.mctq_fq_codes_per_tensor(x.data_ptr(), y.data_ptr(), 72, native.DT_F32, native.CODE_I4, 0.1, 0, -9, 7, st) \
        == native.MCTQ_E_ARG
    assert lib.mctq_fq_codes_per_tensor(x.data_ptr(), y.data_ptr(), 70, native.DT_F32, native.CODE_U4, 0.1, 0, 0, 15, st) \
        == native.MCTQ_E_ARG


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize("shape", [(2, 32, 7, 5), (1, 128, 8, 8), (3, 130, 6, 6), (2, 16, 1, 1), (1, 576, 14, 14), (2, 48, 3, 5)])
def test_fused_quantize_and_nchw_to_nhwc_codes(lib, dtype, shape):
    from mct_quantizers_amd.hip import ops
    from oracle import mctq_oracle as O
    x = (torch.randn(*shape) * 2).to(dtype)
    for (qmin, qmax, scale, zp) in ((0, 255, 0.0219, 114), (-128, 127, 0.031, 0), (0, 15, 0.3, 7)):
        got = ops.fq_codes_nhwc(x.cuda(), qmin, qmax, scale, zp)
        # the ORACLE's clamp index (numpy, from the reference's arithmetic), moved to NHWC on the host
        _, idx = O.fake_quant_affine(x.float().numpy(), [scale], [zp], qmin, qmax, return_index=True)
        want = torch.from_numpy(np.ascontiguousarray(idx.transpose(0, 2, 3, 1))).to(got.dtype)
        assert got.shape == (shape[0], shape[2], shape[3], shape[1]) and got.is_contiguous()
        assert torch.equal(got.cpu(), want), (shape, dtype, qmin)
        want = want.cuda()
        cl = x.cuda().contiguous(memory_format=torch.channels_last)
        assert torch.equal(ops.fq_codes_nhwc(cl, qmin, qmax, scale, zp), want)
        assert torch.equal(ops.fq_codes_nhwc(x, qmin, qmax, scale, zp), want.cpu())          # CPU route
