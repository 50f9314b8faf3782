"""The compact decision table (mctq_lut_build_compact / mctq_lutc_*, include/mctq_hip.h; LutCompactOp in
csrc/mctq_kernels.hpp): same cells and thresholds as the decision table, one byte per cell + the list of steps.
Held to the literal first-minimum scan (the reference's argmin, pytorch/quantizer_utils.py:131-134) for ALL 2^32 float32
inputs per codebook, to the oracle through the raw C ABI over launch shapes / storage types / tuning variants, and to the
decision-table kernels it replaces in the single-tensor launches of the quantizer classes."""
import warnings

import numpy as np
import pytest
import torch

import mct_quantizers_amd as mq
from conftest import bits_equal, finite_equal, first_mismatch

pytestmark = pytest.mark.gpu
Q = mq.pytorch_quantizers

LUTS = {
    "l1": [7.0],
    "l2": [-5.0, 5.0],
    "l3dup": [3.0, 3.0, -8.0],
    "l8": [22.0, -53.0, 62.0, 0.0, -66.0, -21.0, 44.0, -40.0],
    "l16": [-128.0, -96.0, -64.0, -40.0, -24.0, -12.0, -5.0, 0.0, 5.0, 12.0, 24.0, 40.0, 64.0, 96.0, 120.0, 127.0],
    "l40": [float(v) for v in np.random.default_rng(5).permutation(np.arange(-128, 128))[:40]],
    "l256": [float(v) for v in np.random.default_rng(6).permutation(np.arange(-128, 128))],
}


@pytest.fixture(scope="module")
def lib():
    from mct_quantizers_amd.hip import native
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    return native.load()


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _compact(lut, mult=128.0, cmin=-128.0, cmax=127.0):
    from mct_quantizers_amd.hip import native
    blob = native.build_lut_compact(lut, mult, cmin, cmax)
    assert blob is not None
    return _dev(blob)


def _lut_inputs(rng, shape, thr_b):
    x = rng.standard_normal(int(np.prod(shape))).astype(np.float32).reshape(shape) * thr_b * np.float32(0.6)
    mid = (rng.integers(-130, 130, size=shape).astype(np.float32) + np.float32(0.5)) / np.float32(128.0) * thr_b
    x = np.where(rng.integers(0, 4, size=shape) == 0, mid, x).astype(np.float32)
    flat = x.reshape(-1)
    if flat.size >= 8:
        flat[:8] = np.asarray([0.0, -0.0, 1e-9, -1e-9, 1e-39, 3e5, -3e5, 1e30], dtype=np.float32)
    return x


def test_blob_layout_and_size():
    from mct_quantizers_amd.hip import native
    blob = native.build_lut_compact(LUTS["l16"], 128.0, -128.0, 127.0)
    words = blob.view(np.uint32)
    assert words.size == 128 + 2 * 16 + 2 and words.size * 4 == 648                  # 511 cells, 15 steps + terminator, trailer
    assert blob[-1] == 15.0 and blob[-2] == np.float32(-128.0 / 128.0)              # P, q for NaN input = lut[0] / mult
    cells = words[:128].view(np.uint8)
    assert cells[0] == 0 and cells[510] == 15 and np.all(np.diff(cells[:511].astype(int)) >= 0)
    assert np.isinf(blob[128 + 2 * 15]) and blob[128 + 2 * 15] > 0                    # terminator threshold
    assert native.build_lut_compact([0.5, 1.0], 128.0, -128.0, 127.0) is None        # no decision table: no compact table
    many = [float(v) for v in range(-512, 512, 2)]                                   # 511 steps > 255
    assert native.build_lut_compact(many, 512.0, -512.0, 511.0) is None
    assert native.build_lut_table(many, 512.0, -512.0, 511.0) is not None            # ... the full table still serves it


@pytest.mark.parametrize("lut_name", list(LUTS))
def test_compact_table_equals_literal_scan_for_every_float(lib, lut_name):
    """All 2^32 float32 inputs: the compact-table kernel == the literal first-minimum scan kernel."""
    lut = LUTS[lut_name]
    lut_d, blob = _dev(np.asarray(lut, dtype=np.float32)), _compact(lut)
    chunk = 1 << 28
    y_lit = torch.empty(chunk, dtype=torch.float32, device="cuda")
    y_cmp = torch.empty(chunk, dtype=torch.float32, device="cuda")
    for c in range(16):
        bits = torch.arange(c * chunk - (1 << 31), (c + 1) * chunk - (1 << 31), dtype=torch.int64, device="cuda")
        x = bits.to(torch.int32).view(torch.float32)
        del bits
        # thr_div = 1, mult = 128: t = clamp(x * 128) sweeps every float in the clip range
        assert lib.mctq_lut_per_tensor_f32(x.data_ptr(), y_lit.data_ptr(), chunk, 1.0, 1.0, lut_d.data_ptr(),
                                           len(lut), 128.0, -128.0, 127.0, _stream()) == 0
        assert lib.mctq_lutc_per_tensor(x.data_ptr(), y_cmp.data_ptr(), chunk, 0, 0, 1.0, 1.0, blob.data_ptr(),
                                        blob.numel(), 128.0, -128.0, 127.0, _stream()) == 0, lib.mctq_last_error()
        if not torch.equal(y_lit.view(torch.int32), y_cmp.view(torch.int32)):
            i = int(torch.nonzero(y_lit.view(torch.int32) != y_cmp.view(torch.int32))[0])
            raise AssertionError(f"chunk {c}: x={x[i].item()!r} literal={y_lit[i].item()!r} compact={y_cmp[i].item()!r}")
        del x


@pytest.mark.parametrize("divisor", [0.37, 3.0, 1e-3])
def test_compact_table_equals_literal_scan_for_every_float_with_a_real_divisor(lib, divisor):
    """The same sweep through the shared-divisor division (thr_div != 1), 16-entry codebook."""
    lut = LUTS["l16"]
    lut_d, blob = _dev(np.asarray(lut, dtype=np.float32)), _compact(lut)
    chunk = 1 << 28
    y_lit = torch.empty(chunk, dtype=torch.float32, device="cuda")
    y_cmp = torch.empty(chunk, dtype=torch.float32, device="cuda")
    d = float(np.float32(divisor))
    for c in range(16):
        bits = torch.arange(c * chunk - (1 << 31), (c + 1) * chunk - (1 << 31), dtype=torch.int64, device="cuda")
        x = bits.to(torch.int32).view(torch.float32)
        del bits
        assert lib.mctq_lut_per_tensor_f32(x.data_ptr(), y_lit.data_ptr(), chunk, d, d, lut_d.data_ptr(), len(lut), 128.0,
                                           -128.0, 127.0, _stream()) == 0
        assert lib.mctq_lutc_per_tensor(x.data_ptr(), y_cmp.data_ptr(), chunk, 0, 0, d, d, blob.data_ptr(), blob.numel(),
                                        128.0, -128.0, 127.0, _stream()) == 0
        if not torch.equal(y_lit.view(torch.int32), y_cmp.view(torch.int32)):
            i = int(torch.nonzero(y_lit.view(torch.int32) != y_cmp.view(torch.int32))[0])
            raise AssertionError(f"chunk {c}: x={x[i].item()!r} literal={y_lit[i].item()!r} compact={y_cmp[i].item()!r}")
        del x


def test_unsigned_and_wide_codebooks(lib):
    from oracle import mctq_oracle as O
    rng = np.random.default_rng(3)
    cases = [([0.0, 13.0, 50.0, 90.0, 128.0, 200.0, 255.0, 256.0], False, 8),
             ([float(v) for v in range(-512, 512, 64)], True, 10),
             ([-8.0, -3.0, 2.0, 7.0], True, 4)]
    for lut, signed, B in cases:
        mult = float(2 ** (B - int(signed)))
        cmin, cmax = (float(-2 ** (B - 1)), float(2 ** (B - 1) - 1)) if signed else (0.0, float(2 ** B - 1))
        blob = _compact(lut, mult, cmin, cmax)
        thr = 1.7
        x_np = (rng.standard_normal(200003).astype(np.float32) * np.float32(1.2))
        mid = (rng.integers(int(2 * cmin) - 4, int(2 * cmax) + 4, size=x_np.size).astype(np.float32) * np.float32(0.5)
               / np.float32(mult) * np.float32(thr))
        x_np = np.where(rng.integers(0, 3, size=x_np.size) == 0, mid, x_np).astype(np.float32)
        x = _dev(x_np)
        y = torch.empty_like(x)
        thr_div = float(np.float32(thr) + np.float32(1e-8))
        assert lib.mctq_lutc_per_tensor(x.data_ptr(), y.data_ptr(), x.numel(), 0, 0, thr_div, float(np.float32(thr)),
                                        blob.data_ptr(), blob.numel(), mult, cmin, cmax, _stream()) == 0
        want = O.lut_quantize(x_np, lut, np.asarray([thr], dtype=np.float32), signed, B, 1e-8)
        assert bits_equal(y.cpu().numpy(), want), first_mismatch(y.cpu().numpy(), want, x_np)


@pytest.mark.parametrize("lut_name", ["l1", "l3dup", "l16", "l256"])
@pytest.mark.parametrize("outer,C,inner", [(1, 3, 1), (50, 3, 1), (4, 6, 5), (2, 8, 100), (2, 6, 1024), (3, 5, 1028),
                                           (1, 16, 11008), (1, 3000, 3), (1, 2, 70000), (41, 64, 1), (3, 4096, 1),
                                           (1, 7, 4096), (2, 3, 2048), (1, 5, 3072)])
def test_abi_per_channel_vs_oracle(lib, lut_name, outer, C, inner):
    from oracle import mctq_oracle as O
    rng = np.random.default_rng(C * 31 + inner + 1)
    lut = LUTS[lut_name]
    thr = rng.uniform(0.05, 4.0, size=C).astype(np.float32)
    shape = (outer, C, inner)
    x_np = _lut_inputs(rng, shape, thr.reshape(1, C, 1))
    x, t_d, blob = _dev(x_np), _dev(thr), _compact(lut)
    y = torch.empty_like(x)
    rc = lib.mctq_lutc_per_channel(x.data_ptr(), y.data_ptr(), outer, C, inner, 0, t_d.data_ptr(), 1e-8,
                                   blob.data_ptr(), blob.numel(), 128.0, -128.0, 127.0, _stream())
    assert rc == 0, lib.mctq_last_error()
    want = O.lut_quantize(x_np, lut, thr, True, 8, 1e-8, per_channel=True, channel_axis=1)
    assert bits_equal(y.cpu().numpy(), want), first_mismatch(y.cpu().numpy(), want, x_np)


@pytest.mark.parametrize("lut_name", ["l2", "l16", "l40"])
@pytest.mark.parametrize("n,offset", [(0, 0), (5, 0), (4096 + 3, 0), (100000, 1), (1 << 22, 0)])
def test_abi_per_tensor_vs_oracle(lib, lut_name, n, offset):
    from oracle import mctq_oracle as O
    rng = np.random.default_rng(n + len(lut_name))
    lut = LUTS[lut_name]
    thr = 2.0
    x_np = _lut_inputs(rng, (n + offset,), np.float32(thr))
    xb, blob = _dev(x_np), _compact(lut)
    yb = torch.full_like(xb, 777.0)
    thr_div = float(np.float32(thr + 1e-8))
    x, y = xb[offset:], yb[offset:]                          # offset 1: not 16-byte aligned
    rc = lib.mctq_lutc_per_tensor(x.data_ptr(), y.data_ptr(), n, 0, 0, thr_div, thr, blob.data_ptr(), blob.numel(), 128.0,
                                  -128.0, 127.0, _stream())
    assert rc == 0, lib.mctq_last_error()
    want = O.lut_quantize(x_np[offset:], lut, thr, True, 8, 1e-8)
    assert bits_equal(y.cpu().numpy(), want), first_mismatch(y.cpu().numpy(), want, x_np[offset:])
    if offset:
        assert float(yb[0]) == 777.0                         # nothing written in front of y


@pytest.mark.parametrize("dt", ["float16", "bfloat16"])
@pytest.mark.parametrize("outer,C,inner", [(50, 3, 1), (4, 6, 5), (2, 8, 104), (2, 6, 2048), (3, 5, 1032), (1, 16, 11008),
                                           (41, 64, 1), (3, 4096, 1), (1, 3000, 3)])
def test_abi_half_inputs_vs_oracle(lib, dt, outer, C, inner):
    from oracle import mctq_oracle as O
    code = {"float16": 1, "bfloat16": 2}[dt]
    rng = np.random.default_rng(C * 13 + inner)
    lut = LUTS["l16"]
    thr = rng.uniform(0.05, 4.0, size=C).astype(np.float32)
    shape = (outer, C, inner)
    xh = _dev(_lut_inputs(rng, shape, thr.reshape(1, C, 1))).to(getattr(torch, dt))
    x_np = xh.float().cpu().numpy()
    t_d, blob = _dev(thr), _compact(lut)
    y = torch.empty(shape, dtype=torch.float32, device="cuda")
    rc = lib.mctq_lutc_per_channel(xh.data_ptr(), y.data_ptr(), outer, C, inner, code, t_d.data_ptr(), 1e-8,
                                   blob.data_ptr(), blob.numel(), 128.0, -128.0, 127.0, _stream())
    assert rc == 0, lib.mctq_last_error()
    want = O.lut_quantize(x_np, lut, thr, True, 8, 1e-8, per_channel=True, channel_axis=1)
    assert finite_equal(y.cpu().numpy(), want, x_np), first_mismatch(y.cpu().numpy(), want, x_np)
    y1 = torch.empty(xh.numel(), dtype=torch.float32, device="cuda")
    thr_div = float(torch.tensor([2.0 + 1e-8], dtype=torch.float64).to(getattr(torch, dt)).item())
    rc = lib.mctq_lutc_per_tensor(xh.data_ptr(), y1.data_ptr(), xh.numel(), code, code, thr_div, 2.0, blob.data_ptr(),
                                  blob.numel(), 128.0, -128.0, 127.0, _stream())
    assert rc == 0, lib.mctq_last_error()
    want1 = O.lut_quantize(x_np.reshape(-1), lut, 2.0, True, 8, 1e-8, step_dtype=dt)
    assert finite_equal(y1.cpu().numpy(), want1, x_np.reshape(-1)), first_mismatch(y1.cpu().numpy(), want1, x_np.reshape(-1))


def test_tuning_variants_do_not_change_results(lib):
    from mct_quantizers_amd.hip import native
    from oracle import mctq_oracle as O
    rng = np.random.default_rng(12)
    C, inner = 9, 11008
    lut = LUTS["l16"]
    thr = rng.uniform(0.05, 4.0, size=C).astype(np.float32)
    x_np = _lut_inputs(rng, (1, C, inner), thr.reshape(1, C, 1))
    want = O.lut_quantize(x_np, lut, thr, True, 8, 1e-8, per_channel=True, channel_axis=1)
    want1 = O.lut_quantize(x_np, lut, thr[:1], True, 8, 1e-8)
    x, t_d, blob = _dev(x_np), _dev(thr), _compact(lut)
    seen = set()
    try:
        for nt, hu, pers in [(n_, h_, p_) for n_ in (0, 1, 2) for h_ in (1, 2, 4, 8, 0) for p_ in (0, 1)]:
            native.set_tuning("nt", nt)
            native.set_tuning("heavy_unroll", hu)
            native.set_tuning("heavy_persistent", pers)
            y = torch.empty_like(x)
            rc = lib.mctq_lutc_per_channel(x.data_ptr(), y.data_ptr(), 1, C, inner, 0, t_d.data_ptr(), 1e-8,
                                           blob.data_ptr(), blob.numel(), 128.0, -128.0, 127.0, _stream())
            assert rc == 0, lib.mctq_last_error()
            seen.add(native.last_launch())
            assert bits_equal(y.cpu().numpy(), want), (nt, hu, pers)
            y = torch.empty_like(x)
            rc = lib.mctq_lutc_per_tensor(x.data_ptr(), y.data_ptr(), x.numel(), 0, 0, float(thr[0] + np.float32(1e-8)),
                                          float(thr[0]), blob.data_ptr(), blob.numel(), 128.0, -128.0, 127.0, _stream())
            assert rc == 0, lib.mctq_last_error()
            assert bits_equal(y.cpu().numpy(), want1), (nt, hu, pers, "per-tensor")
    finally:
        native.set_tuning("nt", 1)
        native.set_tuning("heavy_unroll", 0)
        native.set_tuning("heavy_persistent", 0)
    assert any("rows_kernel<LutCompactOp" in s and "U=1" in s for s in seen) and any("U=4" in s for s in seen), seen


def test_quantizer_classes_take_the_compact_table_and_agree_with_the_full_one(lib, monkeypatch):
    from mct_quantizers_amd.hip import native, ops
    from oracle import oracle_call
    rng = np.random.default_rng(4)
    x_np = (rng.standard_normal((64, 11008)) * 0.8).astype(np.float32)
    thr = [float(v) for v in np.abs(x_np).max(axis=1)]
    kw = dict(num_bits=4, lut_values=LUTS["l16"], threshold=thr, per_channel=True, channel_axis=0, input_rank=2)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        want = oracle_call("WeightsLUTSymmetricInferableQuantizer", kw, x_np)
    x = _dev(x_np)
    monkeypatch.setattr(ops, "USE_COMPACT_LUT", True)                      # what MCTQ_COMPACT_LUT=1 sets at import
    q = Q.WeightsLUTSymmetricInferableQuantizer(**kw)
    assert getattr(q._lut_table_torch, "_mctq_compact", None) is not None
    y = q(x)
    assert "LutCompactOp" in native.last_launch(), native.last_launch()
    assert bits_equal(y.cpu().numpy(), want), first_mismatch(y.cpu().numpy(), want, x_np)
    monkeypatch.setattr(ops, "USE_COMPACT_LUT", False)                     # the default
    q2 = Q.WeightsLUTSymmetricInferableQuantizer(**kw)
    assert getattr(q2._lut_table_torch, "_mctq_compact", None) is None
    y2 = q2(x)
    assert "LutTableOp" in native.last_launch(), native.last_launch()
    assert torch.equal(y, y2)
    monkeypatch.setattr(ops, "USE_COMPACT_LUT", True)
    # per-tensor activation quantizer (pre-packed LutPlan) and half-precision activations
    kwa = dict(num_bits=4, lut_values=LUTS["l16"], threshold=[4.0], signed=True)
    qa = Q.ActivationLutPOTInferableQuantizer(**kwa)
    xa_np = (rng.standard_normal((2, 3, 32, 32)) * 2).astype(np.float32)
    ya = qa(_dev(xa_np))
    assert "LutCompactOp" in native.last_launch(), native.last_launch()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        assert bits_equal(ya.cpu().numpy(), oracle_call("ActivationLutPOTInferableQuantizer", kwa, xa_np))
        xh = _dev(xa_np).half()
        got = qa(xh).cpu().numpy()
        wanth = oracle_call("ActivationLutPOTInferableQuantizer", kwa, xh.float().cpu().numpy(), in_dtype="float16")
    assert bits_equal(got, wanth), first_mismatch(got, wanth)
