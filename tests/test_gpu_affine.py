"""GPU parity of the affine quantizers beyond the golden / raw-ABI cases of test_gpu_parity.py: float64 tensors (ATen's
double arithmetic), the tensor-qparams entry point, every 16-bit input value, config 3 at every batch size against the
reference's digests, fuzz over bit widths / signs / classes against ATen's CPU operator, launch-state invalidation when a
public attribute is assigned, ATen's error behaviour.  All against the oracle / reference fixtures / ATen CPU."""
import hashlib
import json
import os
import warnings

import numpy as np
import pytest
import torch

from conftest import GOLDEN, bits_equal, finite_equal, first_mismatch, load_json

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    from mct_quantizers_amd.hip import native
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    return native.load()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint64 if a.dtype == np.float64 else np.uint32)


def _make(cls, kwargs):
    import mct_quantizers_amd as mq
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return getattr(mq.pytorch_quantizers, cls)(**kwargs)


# ---------------------------------------------------------------------------------------------
# float64
# ---------------------------------------------------------------------------------------------

def test_float64_golden_cases_via_quantizer_classes(lib):
    """41 (kwargs, float64 input) -> output cases produced by the reference: dtype, strides and every bit."""
    meta = load_json("cases_f64.json")
    arrays = np.load(os.path.join(GOLDEN, "cases_f64.npz"))
    assert len(meta["cases"]) >= 40
    for c in meta["cases"]:
        x_np, want = arrays[c["id"] + "_x"], arrays[c["id"] + "_y"]
        x = _dev(x_np)
        if c["memory_format"] == "channels_last":
            x = x.contiguous(memory_format=torch.channels_last)
        y = _make(c["cls"], c["kwargs"])(x)
        assert y.is_cuda and y.shape == x.shape and str(y.dtype) == "torch." + c["out_dtype"], c["id"]
        got = y.cpu().numpy()
        assert got.dtype == want.dtype and np.array_equal(_bits(got), _bits(want)), \
            f'{c["id"]} {c["cls"]} {c["shape"]}: {(got != want).sum()} of {got.size} differ'


def test_float64_raw_abi_against_oracle_all_layouts(lib):
    from mct_quantizers_amd.hip import native
    from oracle import mctq_oracle as O
    rng = np.random.default_rng(5)
    st = torch.cuda.current_stream().cuda_stream
    for outer, C, inner in ((1, 1, 4099), (3, 5, 7), (2, 6, 64), (1, 4, 2048), (5, 3, 1), (1, 1, 1), (2, 2, 3)):
        n = outer * C * inner
        s = rng.uniform(0.01, 0.2, size=C).astype(np.float32)
        z = rng.integers(-4, 5, size=C).astype(np.int32)
        x = rng.standard_normal(n) * 3.0
        sd, zd = _dev(s), _dev(z)                                # keep the device tables alive across the launch
        for offset in (0, 1):                                   # 16-byte aligned and not
            xb = torch.zeros(n + 2, dtype=torch.float64, device="cuda")
            xb[offset:offset + n] = _dev(x)
            yb = torch.zeros(n + 2, dtype=torch.float64, device="cuda")
            rc = lib.mctq_fq_per_channel(xb[offset:].data_ptr(), yb[offset:].data_ptr(), outer, C, inner, native.DT_F64,
                                         sd.data_ptr(), zd.data_ptr(), -8, 7, st)
            assert rc == 0, native.load().mctq_last_error()
            want = O.fake_quant_affine_f64(x.reshape(outer, C, inner), s, z, -8, 7, axis=1).reshape(-1)
            got = yb[offset:offset + n].cpu().numpy()
            assert np.array_equal(_bits(got), _bits(want)), (outer, C, inner, offset)
            assert float(yb[offset + n:].abs().sum()) == 0 and float(yb[:offset].abs().sum()) == 0   # no stray writes
    # per tensor: float qparams and device qparams give the float32-product flavour
    x = rng.standard_normal(1000) * 3.0
    xd, yd = _dev(x), torch.empty(1000, dtype=torch.float64, device="cuda")
    want = O.fake_quant_affine_f64(x, [0.0371], [3], 0, 255)
    assert lib.mctq_fq_per_tensor(xd.data_ptr(), yd.data_ptr(), 1000, native.DT_F64, 0.0371, 3, 0, 255, st) == 0
    assert np.array_equal(_bits(yd.cpu().numpy()), _bits(want))
    sc, zp = _dev(np.float32([0.0371])), _dev(np.int32([3]))
    yd.zero_()
    assert lib.mctq_fq_per_tensor_tqp(xd.data_ptr(), yd.data_ptr(), 1000, native.DT_F64, sc.data_ptr(), zp.data_ptr(), 0, 255, st) == 0
    assert np.array_equal(_bits(yd.cpu().numpy()), _bits(want))


# ---------------------------------------------------------------------------------------------
# tensor-qparams entry point, fx routing of traced reference quantizers
# ---------------------------------------------------------------------------------------------

def test_tensor_qparams_entry_against_oracle(lib):
    from mct_quantizers_amd.hip import ops
    from oracle import mctq_oracle as O
    rng = np.random.default_rng(31)
    for n in (1, 3, 1023, 1024, 4096 + 5, 1 << 20):
        x_np = (rng.standard_normal(n) * 2).astype(np.float32)
        s, z = np.float32([0.0173]), np.int32([-2])
        want = O.fake_quant_affine(x_np, s, z, -128, 127)
        for dt in (torch.float32, torch.float16, torch.bfloat16):
            x = torch.from_numpy(x_np).to(dt).cuda()
            w = O.narrow(O.fake_quant_affine(x.float().cpu().numpy(), s, z, -128, 127), str(dt).replace("torch.", ""))
            for got in (ops.fq_per_tensor_tqp(x, _dev(s), _dev(z), -128, 127),
                        ops._hip_fq_per_tensor_tqp(x, _dev(s), _dev(z), -128, 127),
                        torch.ops.mctq_amd.fq_per_tensor_tqp(x, _dev(s), _dev(z), -128, 127)):
                assert got.dtype == dt and bits_equal(got.float().cpu().numpy(), w), (n, dt)
        assert bits_equal(ops.fq_per_tensor_tqp(_dev(x_np), _dev(s), _dev(z), -128, 127).cpu().numpy(), want)
    with pytest.raises(RuntimeError):
        ops.fq_per_tensor_tqp(_dev(np.float32([1, 2])), torch.tensor([0.1]), _dev(np.int32([0])), -8, 7)   # CPU scale


# ---------------------------------------------------------------------------------------------
# config 3 at the other batch sizes SURVEY §8(d) names
# ---------------------------------------------------------------------------------------------

@pytest.mark.parametrize("n", [1, 64, 256])
def test_config3_full_size_digests_for_every_batch_size(lib, n):
    from mct_quantizers_amd import workloads
    rec = load_json("full_sha.json")["configs"][f"cfg3_n{n}"]
    x_np = workloads.make_input("cfg3", batch=n)
    assert hashlib.sha256(x_np.tobytes()).hexdigest() == rec["x_sha256"]
    wl = workloads.make_workload("cfg3", x_np)
    y = _make(wl.quantizer, wl.kwargs)(_dev(x_np)).cpu().numpy()
    assert hashlib.sha256(np.ascontiguousarray(y).tobytes()).hexdigest() == rec["y_sha256"]


# ---------------------------------------------------------------------------------------------
# launch state follows the public attributes (the reference reads them on every call)
# ---------------------------------------------------------------------------------------------

def test_public_attribute_changes_take_effect(lib):
    import mct_quantizers_amd as mq
    Q = mq.pytorch_quantizers
    x = torch.randn(3, 64, device="cuda")
    q = Q.ActivationUniformInferableQuantizer(8, [-2.5], [3.1])
    a = q(x)
    q.scale = q.scale * 2
    b = q(x)
    assert torch.equal(b, torch.fake_quantize_per_tensor_affine(x, q.scale, q.zero_point, 0, 255)) and not torch.equal(a, b)
    q.zero_point = 100
    assert torch.equal(q(x), torch.fake_quantize_per_tensor_affine(x, q.scale, 100, 0, 255))
    qs = Q.ActivationSymmetricInferableQuantizer(8, [2.0], True)
    qs.scales = 0.05
    assert torch.equal(qs(x), torch.fake_quantize_per_tensor_affine(x, 0.05, 0, -128, 127))
    w = torch.randn(3, 64, device="cuda")
    qw = Q.WeightsSymmetricInferableQuantizer(8, [1.0, 2.0, 3.0], True, 0)
    base = qw(w.clone())
    qw.zero_points[0] = 5                                           # in-place edit of the device tensor
    got = qw(w.clone())
    want = torch.fake_quantize_per_channel_affine(w, qw.scales, qw.zero_points, 0, -128, 127)
    assert torch.equal(got, want) and not torch.equal(got, base)
    qw.scales = qw.scales * 0.5                                     # replaced tensor
    assert torch.equal(qw(w.clone()), torch.fake_quantize_per_channel_affine(w, qw.scales, qw.zero_points, 0, -128, 127))
    qt = Q.WeightsUniformInferableQuantizer(8, [-1.0], [1.0], False)
    qt.scales.mul_(2.0)
    assert torch.equal(qt(w.clone()), torch.fake_quantize_per_tensor_affine(w, qt.scales, qt.zero_points, 0, 255))
    import copy
    import pickle
    for obj in (q, qs, qw, qt):
        clone = pickle.loads(pickle.dumps(obj))
        assert torch.equal(clone(w.clone()), obj(w.clone())) and torch.equal(copy.deepcopy(obj)(w.clone()), obj(w.clone()))
    # clamp domains beyond the kernels' float32 bounds are accepted, as in the reference, and run ATen's operator on the GPU
    wide = Q.ActivationSymmetricInferableQuantizer(30, [2.0], True)
    big = torch.randn(3, 64, device="cuda") * 1e5
    assert torch.equal(wide(big), torch.fake_quantize_per_tensor_affine(big, wide.scales, 0, -2 ** 29, 2 ** 29 - 1))
    ww = Q.WeightsSymmetricInferableQuantizer(28, [1.0, 2.0, 3.0], True, 0)
    assert torch.equal(ww(big.clone()), torch.fake_quantize_per_channel_affine(big, ww.scales, ww.zero_points, 0, -2 ** 27, 2 ** 27 - 1))


def test_parameters_on_another_device_raise_cleanly(lib):
    from mct_quantizers_amd.hip import ops
    x = torch.randn(4, 8, device="cuda")
    with pytest.raises(RuntimeError, match="same device"):
        ops.fq_per_channel(x, torch.ones(4), torch.zeros(4, dtype=torch.int32), 0, -8, 7)
    with pytest.raises(RuntimeError, match="same device"):
        ops.fq_codes(x, torch.ones(4), torch.zeros(4, dtype=torch.int32), 0, -8, 7)
    with pytest.raises(RuntimeError, match="same device"):
        ops.lut_per_channel(x, torch.tensor([0.0, 1.0]), torch.ones(4, device="cuda"), 1e-8, 0, 128.0, -128.0, 127.0)


def test_fuzz_float64_tensor_qparams_and_batched_against_aten_cpu(lib):
    """Seeded fuzz over ranks, shapes, axes, permuted storage and ALL FOUR storage types (incl. float64), for the
    per-channel quantizers, the per-tensor weights quantizers (tensor qparams) and the activation quantizers; every
    dense case is also pushed through the batched launch and must give the same bits.  Reference: ATen's CPU
    operators on the same tensor (what the reference package executes)."""
    import mct_quantizers_amd as mq
    from mct_quantizers_amd.hip import ops
    Q = mq.pytorch_quantizers
    rng = np.random.default_rng(int(os.environ.get("MCTQ_FUZZ_SEED", "77")))
    pending = []
    for case in range(int(os.environ.get("MCTQ_FUZZ_CASES", "140"))):
        rank = int(rng.integers(1, 5))
        shape = [int(rng.choice([1, 2, 3, 4, 5, 7, 8, 16, 33, 64])) for _ in range(rank)]
        if rng.random() < 0.3:
            shape[int(rng.integers(0, rank))] = int(rng.choice([257, 1024, 1030, 4096, 8192]))
        if int(np.prod(shape, dtype=np.int64)) > (1 << 24):
            continue
        axis = int(rng.integers(0, rank))
        dt = [torch.float32, torch.float64, torch.float64, torch.float16, torch.bfloat16][int(rng.integers(0, 5))]
        bits = int(rng.choice([2, 4, 8]))
        x = torch.from_numpy(rng.standard_normal(shape) * 3).to(dt)
        perm = list(rng.permutation(rank))
        x = x.permute(perm).contiguous().permute(list(np.argsort(perm)))
        C = x.shape[axis]
        kind = int(rng.integers(0, 4))
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            if kind == 0:
                q = Q.WeightsSymmetricInferableQuantizer(bits, [float(v) for v in rng.uniform(0.2, 6.0, size=C)], True, axis)
                ref = lambda t, q=q, axis=axis: torch.fake_quantize_per_channel_affine(   # noqa: E731
                    t, q.scales.cpu(), q.zero_points.cpu(), axis, q.min_quantized_domain, q.max_quantized_domain)
            elif kind == 1:
                lo = [float(v) for v in rng.uniform(-4.0, -0.1, size=C)]
                hi = [float(v) for v in rng.uniform(0.1, 5.0, size=C)]
                q = Q.WeightsUniformInferableQuantizer(bits, lo, hi, True, axis)
                ref = lambda t, q=q, axis=axis, bits=bits: torch.fake_quantize_per_channel_affine(   # noqa: E731
                    t, q.scales.cpu(), q.zero_points.cpu(), axis, 0, 2 ** bits - 1)
            elif kind == 2:
                q = Q.WeightsUniformInferableQuantizer(bits, [float(rng.uniform(-3, -0.1))], [float(rng.uniform(0.1, 4))], False)
                ref = lambda t, q=q, bits=bits: torch.fake_quantize_per_tensor_affine(   # noqa: E731  (tensor qparams)
                    t, q.scales.cpu(), q.zero_points.cpu(), 0, 2 ** bits - 1)
            else:
                q = Q.ActivationSymmetricInferableQuantizer(bits, [float(rng.uniform(0.5, 5))], bool(rng.integers(0, 2)))
                ref = lambda t, q=q: torch.fake_quantize_per_tensor_affine(   # noqa: E731
                    t, q.scales, q.zero_points, q.min_quantized_domain, q.max_quantized_domain)
        want = ref(x.clone())
        xg = x.cuda()
        got = q(xg)
        info = (case, tuple(x.shape), x.stride(), axis, dt, kind)
        view = torch.int64 if dt == torch.float64 else torch.int32
        conv = (lambda t: t) if dt == torch.float64 else (lambda t: t.float())
        assert got.dtype == want.dtype and got.shape == want.shape and got.stride() == xg.stride(), info
        assert torch.equal(conv(got.cpu()).contiguous().view(view), conv(want).contiguous().view(view)), info
        if kind < 3:
            pending.append((q.batch_item(xg), got, info))
        if len(pending) >= 9 or (pending and case % 37 == 36):
            outs = ops.fq_batched([p[0] for p in pending])
            for y, (_, single, inf) in zip(outs, pending):
                assert y.dtype == single.dtype and y.stride() == single.stride() and torch.equal(y, single), ("batched", inf)
            pending = []


def test_per_tensor_argument_errors_match_aten(lib):
    """ATen validates the host-known per-tensor qparams before launching; same exception type and message here,
    through both bindings."""
    from mct_quantizers_amd.hip import native, ops
    x = torch.randn(16, device="cuda")
    for args in ((0.1, 300, 0, 255), (0.1, -1, 0, 255), (0.1, 0, 5, 3)):
        with pytest.raises(RuntimeError) as want:
            torch.fake_quantize_per_tensor_affine(x, *args)
        msg = str(want.value).splitlines()[0]
        fns = [ops.fq_per_tensor, ops._hip_fq_per_tensor] + ([native.fast().fq_per_tensor] if native.fast() is not None else [])
        for f in fns:
            with pytest.raises(RuntimeError) as got:
                f(x, *args)
            assert str(got.value).splitlines()[0] == msg, (args, f)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_every_16_bit_input_value_affine_and_lut(lib, dtype):
    """ALL 65 536 bit patterns of a float16 / bfloat16 tensor (finite ones inside the parity domain, plus NaN / inf
    handling checked separately) through the per-tensor and per-channel affine kernels against ATen's CPU operators,
    and through the LUT quantizers (decision table AND literal scan, weights and activation flavours) against the
    oracle."""
    import mct_quantizers_amd as mq
    from mct_quantizers_amd.hip import native, ops
    from oracle import mctq_oracle as O
    Q = mq.pytorch_quantizers
    name = str(dtype).replace("torch.", "")
    bits = torch.arange(65536, dtype=torch.int32).to(torch.int16)
    x = bits.view(dtype)
    finite = torch.isfinite(x.float())
    # affine: parity domain |x / s| < 2^31 -> keep every finite value whose quotient is in range for the scale used
    for scale, zp, qmin, qmax in ((0.0371, 17, 0, 255), (2.0 ** -7, 0, -128, 127), (3.0, -2, -8, 7)):
        ok = finite & ((x.float().abs() / scale) < 2.0 ** 31)
        xs = x[ok]
        want = torch.fake_quantize_per_tensor_affine(xs, scale, zp, qmin, qmax)
        got = ops.fq_per_tensor(xs.cuda(), scale, zp, qmin, qmax)
        assert got.dtype == dtype and torch.equal(got.cpu().view(torch.int16), want.view(torch.int16)), (name, scale)
        # per channel: the same values laid out as 3 channels with different scales
        xc = xs[: (xs.numel() // 24) * 24].reshape(3, -1)
        sc = torch.tensor([scale, scale * 1.7, scale * 0.31], dtype=torch.float32)
        zc = torch.tensor([zp, qmin, qmax], dtype=torch.int32)
        okc = (xc.float().abs() / sc[:, None]) < 2.0 ** 31
        xc = torch.where(okc, xc, torch.zeros_like(xc))
        want = torch.fake_quantize_per_channel_affine(xc, sc, zc, 0, qmin, qmax)
        got = ops.fq_per_channel(xc.cuda(), sc.cuda(), zc.cuda(), 0, qmin, qmax)
        assert torch.equal(got.cpu().view(torch.int16), want.view(torch.int16)), (name, scale, "per channel")
    # saturation outside the domain: +inf -> qmax, -inf / NaN -> qmin (documented divergence from the CPU's UB)
    special = torch.tensor([float("inf"), float("-inf"), float("nan")]).to(dtype).cuda()
    y = ops.fq_per_tensor(special, 0.5, 0, -8, 7).float().cpu()
    assert y.tolist() == [3.5, -4.0, -4.0]
    # LUT: every finite value; decision table and literal scan; activation (Python-float threshold, per-step rounding
    # in the tensor's type) and weights (float32 tensor threshold, promoted chain)
    lut = [-128.0, -96.0, -64.0, -40.0, -24.0, -12.0, -5.0, 0.0, 5.0, 12.0, 24.0, 40.0, 64.0, 96.0, 120.0, 127.0]
    xf = x[finite]
    xw = xf.float().numpy()
    code = native.DT_F16 if dtype == torch.float16 else native.DT_BF16
    lut_d = torch.tensor(lut, device="cuda")
    for thr in (2.0, 0.5):
        qa = Q.ActivationLutPOTInferableQuantizer(4, lut, [thr], True)
        want = O.lut_quantize(xw, lut, thr, True, 8, 1e-8, step_dtype=name)
        got = qa(xf.cuda())
        assert got.dtype == torch.float32 and bits_equal(got.cpu().numpy(), want), (name, thr, "table", first_mismatch(got.cpu().numpy(), want, xw))
        div = float(torch.tensor([thr + 1e-8], dtype=torch.float64).to(dtype).item())
        got = ops._hip_lut_per_tensor(xf.cuda(), lut_d, div, thr, 128.0, -128.0, 127.0, None, code)
        assert bits_equal(got.cpu().numpy(), want), (name, thr, "scan", first_mismatch(got.cpu().numpy(), want, xw))
        qw = Q.WeightsLUTSymmetricInferableQuantizer(4, lut, [thr], False)
        want = O.lut_quantize(xw, lut, np.float32([thr]), True, 8, 1e-8)
        got = qw(xf.cuda())
        assert bits_equal(got.cpu().numpy(), want), (name, thr, "weights", first_mismatch(got.cpu().numpy(), want, xw))


def test_float64_large_random_and_tie_inputs_against_aten_cpu(lib):
    """2^22 doubles per case -- random mantissas over 40 binades, exact ties of the double product and their one-ulp
    neighbours -- through the float64 kernels against ATen's CPU operators (per tensor with float and with tensor
    qparams, per channel along both axes)."""
    from mct_quantizers_amd.hip import ops
    rng = np.random.default_rng(97)
    n = 1 << 22
    for scale, zp, qmin, qmax in ((0.0371, 17, 0, 255), (2.0 ** -7, 0, -128, 127), (1.0 / 3.0, -3, -8, 7)):
        sf = np.float32(scale)
        inv = np.float64(np.float32(1.0) / sf)
        x = rng.standard_normal(n) * np.exp2(rng.integers(-20, 20, size=n))
        k = rng.integers(qmin - 4, qmax + 5, size=n // 4).astype(np.float64) - zp + 0.5
        ties = k / inv
        x[: n // 4] = np.where(rng.random(n // 4) < 0.34, ties, np.where(rng.random(n // 4) < 0.5, np.nextafter(ties, np.inf), np.nextafter(ties, -np.inf)))
        x = x[np.abs(x * inv) < 2.0 ** 31]
        xt = torch.from_numpy(x)
        want = torch.fake_quantize_per_tensor_affine(xt, float(sf), zp, qmin, qmax)
        got = ops.fq_per_tensor(xt.cuda(), float(sf), zp, qmin, qmax)
        assert got.dtype == torch.float64 and torch.equal(got.cpu().view(torch.int64), want.view(torch.int64)), scale
        st, zt = torch.tensor([sf]), torch.tensor([zp], dtype=torch.int32)
        want = torch.fake_quantize_per_tensor_affine(xt, st, zt, qmin, qmax)
        got = ops.fq_per_tensor_tqp(xt.cuda(), st.cuda(), zt.cuda(), qmin, qmax)
        assert torch.equal(got.cpu().view(torch.int64), want.view(torch.int64)), (scale, "tensor qparams")
        m = (x.size // 96) * 96
        for shape, axis in (((3, m // 3), 0), ((m // 32, 32), 1)):
            xc = xt[:m].reshape(shape)
            C = shape[axis]
            sc = torch.from_numpy((sf * rng.uniform(0.5, 2.0, size=C)).astype(np.float32))
            zc = torch.from_numpy(rng.integers(qmin, qmax + 1, size=C).astype(np.int32))
            bs = [1, 1]; bs[axis] = -1
            okc = (xc.abs() * (1.0 / sc.double()).reshape(bs)) < 2.0 ** 31
            xc = torch.where(okc, xc, torch.zeros_like(xc))
            want = torch.fake_quantize_per_channel_affine(xc, sc, zc, axis, qmin, qmax)
            got = ops.fq_per_channel(xc.cuda(), sc.cuda(), zc.cuda(), axis, qmin, qmax)
            assert torch.equal(got.cpu().view(torch.int64), want.view(torch.int64)), (scale, shape, axis)


def test_versioned_reuse_is_not_fooled_by_a_recycled_address(lib):
    """The caching allocator hands a freed block to the next tensor of that size: same address, same shape, same
    version counter, different values.  The cache must key on the tensor OBJECT."""
    import mct_quantizers_amd as mq
    q = mq.pytorch_quantizers.WeightsSymmetricInferableQuantizer(8, [1.0, 2.0, 3.0, 4.0], True, 0)
    q.enable_versioned_reuse()
    a = torch.randn(4, 1024, device="cuda")
    ptr = a.data_ptr()
    ya = q(a)
    assert q(a) is ya
    del a
    b = torch.randn(4, 1024, device="cuda")
    if b.data_ptr() != ptr:
        pytest.skip("the allocator did not recycle the block")
    yb = q(b)
    assert yb is not ya and torch.equal(yb, torch.fake_quantize_per_channel_affine(b, q.scales, q.zero_points, 0, -128, 127))


def test_fuzz_affine_bit_widths_signs_and_every_class_against_aten_cpu(lib):
    """Seeded fuzz over num_bits 1..16, all six affine classes (per tensor and per channel, signed and unsigned), storage
    types and permuted layouts: the HIP result equals ATen's CPU operator on the same tensor with the same parameters."""
    import mct_quantizers_amd as mq
    Q = mq.pytorch_quantizers
    rng = np.random.default_rng(int(os.environ.get("MCTQ_FUZZ_SEED", "99")))
    for case in range(int(os.environ.get("MCTQ_FUZZ_CASES", "150"))):
        rank = int(rng.integers(1, 5))
        shape = [int(rng.choice([1, 2, 3, 5, 8, 16, 33, 64])) for _ in range(rank)]
        if rng.random() < 0.25:
            shape[int(rng.integers(0, rank))] = int(rng.choice([257, 1024, 1030, 4096]))
        if int(np.prod(shape, dtype=np.int64)) > (1 << 22):
            continue
        axis = int(rng.integers(0, rank))
        C = shape[axis]
        bits = int(rng.choice([1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 16]))
        dt = [torch.float32, torch.float32, torch.float16, torch.bfloat16, torch.float64][int(rng.integers(0, 5))]
        scale_mag = float(rng.choice([1e-3, 0.1, 1.0, 30.0]))
        x = (torch.from_numpy(rng.standard_normal(shape).astype(np.float32)) * scale_mag * 2).to(dt)
        perm = list(rng.permutation(rank))
        x = x.permute(perm).contiguous().permute(list(np.argsort(perm)))
        kind = int(rng.integers(0, 8))
        pc = bool(rng.integers(0, 2))
        n = C if pc else 1
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            if kind == 0:
                q = Q.WeightsSymmetricInferableQuantizer(bits, [float(v) for v in rng.uniform(0.2, 4.0, n) * scale_mag], pc, axis if pc else None)
            elif kind == 1:
                q = Q.WeightsPOTInferableQuantizer(bits, [float(2.0 ** e) for e in rng.integers(-6, 5, n)], pc, axis if pc else None)
            elif kind == 2:
                lo = [float(v) for v in rng.uniform(-4.0, 0.5, n) * scale_mag]
                hi = [float(a + d) for a, d in zip(lo, rng.uniform(0.2, 6.0, n) * scale_mag)]
                q = Q.WeightsUniformInferableQuantizer(bits, lo, hi, pc, axis if pc else None)
            elif kind == 3:
                q = Q.ActivationSymmetricInferableQuantizer(bits, [float(rng.uniform(0.2, 4.0) * scale_mag)], bool(rng.integers(0, 2)))
            elif kind == 4:
                q = Q.ActivationPOTInferableQuantizer(bits, [float(2.0 ** rng.integers(-6, 5))], bool(rng.integers(0, 2)))
            else:
                lo = float(rng.uniform(-4.0, 0.5) * scale_mag)
                q = Q.ActivationUniformInferableQuantizer(bits, [lo], [lo + float(rng.uniform(0.2, 6.0) * scale_mag)])
        if kind <= 2:
            s, z = q.scales.cpu(), q.zero_points.cpu()
            if pc:
                ref = torch.fake_quantize_per_channel_affine(x.clone(), s, z, axis, q.min_quantized_domain, q.max_quantized_domain)
            else:
                ref = torch.fake_quantize_per_tensor_affine(x.clone(), s, z, q.min_quantized_domain, q.max_quantized_domain)
        elif kind <= 4:
            ref = torch.fake_quantize_per_tensor_affine(x.clone(), q.scales, q.zero_points, q.min_quantized_domain, q.max_quantized_domain)
        else:
            ref = torch.fake_quantize_per_tensor_affine(x.clone(), q.scale, q.zero_point, q.min_quantized_domain, q.max_quantized_domain)
        got = q(x.cuda())
        info = (case, tuple(x.shape), x.stride(), axis, dt, kind, bits, pc)
        assert got.dtype == ref.dtype and got.shape == ref.shape and got.stride() == ref.stride(), info
        same = torch.equal(got.cpu().double().view(torch.int64), ref.double().view(torch.int64))
        assert same, (info, first_mismatch(got.cpu().double().numpy(), ref.double().numpy(), x.double().numpy()))


def test_error_behaviour_follows_aten_for_axis_zero_points_and_tensor_qparams(lib):
    """What ATen raises (type and message) for an axis out of range, a negative axis and out-of-range per-channel zero
    points, and what it accepts: tensor qparams longer than one element (element 0 is used)."""
    import mct_quantizers_amd as mq
    from mct_quantizers_amd.hip import ops
    Q = mq.pytorch_quantizers
    x = torch.randn(4, 3, device="cuda")
    s, z = torch.tensor([0.1, 0.2, 0.3], device="cuda"), torch.zeros(3, dtype=torch.int32, device="cuda")
    for fn in (lambda a: torch.fake_quantize_per_channel_affine(x, s, z, a, -128, 127), lambda a: ops.fq_per_channel(x, s, z, a, -128, 127)):
        with pytest.raises(IndexError, match=r"Dimension out of range \(expected to be in range of \[-2, 1\], but got 2\)"):
            fn(2)
        with pytest.raises(RuntimeError, match="`axis` must be between 0 and number of dimensions of input"):
            fn(-1)
    want = torch.fake_quantize_per_tensor_affine(x, s[:2], z[:2], -128, 127)            # ATen reads element 0
    assert torch.equal(ops.fq_per_tensor_tqp(x, s[:2], z[:2], -128, 127), want)
    q = Q.WeightsUniformInferableQuantizer(8, [-1.0, -2.0, -0.5], [1.0, 1.0, 2.0], True, 1)
    q(x.clone())
    q.zero_points = torch.tensor([0, 300, 0], dtype=torch.int32, device="cuda")
    with pytest.raises(RuntimeError, match="`zero_point` must be between `quant_min` and `quant_max`."):
        q(x.clone())
    with pytest.raises(RuntimeError, match="`zero_point` must be between `quant_min` and `quant_max`."):
        torch.fake_quantize_per_channel_affine(x, q.scales, q.zero_points, 1, 0, 255)


# ---------------------------------------------------------------------------------------------
# round 6: the channel-last launch (slab grid, exact five-instruction reciprocal, no zero-point table) and the per-lane-vector
# launch of short / ragged rows in every storage type
# ---------------------------------------------------------------------------------------------

def test_fast_reciprocal_is_exact(lib):
    """recip_exact (csrc/mctq_kernels.hpp) == the compiler's IEEE 1.0f / d for EVERY float32 bit pattern of its range."""
    out = torch.zeros(3, dtype=torch.int64, device="cuda")
    assert lib.mctq_selftest_reciprocal(out.data_ptr(), _stream()) == 0, lib.mctq_last_error()
    torch.cuda.synchronize()
    seen, bad, first = [int(v) for v in out.cpu()]
    assert seen == 2 * (0x71800000 - 0x0d800000 + 1) and bad == 0, (seen, bad, hex(max(first - 1, 0)))


R6_SHAPES = [  # (outer, C, inner): lastaxis (inner 1, C % N == 0) ...
    (37, 64, 1), (4099, 4096, 1), (5, 4096, 1), (1029, 200, 1), (70000, 16, 1), (3, 20480, 1), (2051, 768, 1), (1, 8, 1),
    # ... short and ragged rows (gather / window): whole vectors, crossing vectors, rows longer than a tile, wrapping channels
    (1, 4000, 16), (1, 4000, 20), (3, 50, 1020), (2, 7, 4100), (1, 3, 4099), (5, 33, 12), (2, 1000, 40), (1, 2000, 64), (7, 9, 8),
    (1, 300, 576), (4, 2, 9000)]


@pytest.mark.parametrize("dt", ["float32", "float16", "bfloat16"])
@pytest.mark.parametrize("with_zp", [False, True])
def test_channel_last_and_short_row_launches_vs_oracle(lib, dt, with_zp):
    """Every shape x {symmetric (NULL zero-point table), zero points} x the three routes of 16-bit short rows (tuning key
    "shortrows"), scales that include values outside recip_exact's range (the wave falls back to the IEEE division) -- against
    the oracle, bit for bit."""
    from mct_quantizers_amd.hip import native
    from oracle import mctq_oracle as O
    code = {"float32": 0, "float16": 1, "bfloat16": 2}[dt]
    tdt = getattr(torch, dt)
    seen = set()
    try:
        for route in ((1,) if dt == "float32" else (0, 1, 2)):
            native.set_tuning("shortrows", route)
            for k, (outer, C, inner) in enumerate(R6_SHAPES):
                rng = np.random.default_rng(1000 * k + 7 * route + with_zp)
                qmin, qmax = (-8, 7) if k % 3 == 0 else (-128, 127)
                scales = rng.uniform(0.01, 0.2, size=C).astype(np.float32)
                if k % 4 == 1:                                   # one channel per wave-sized stretch leaves the exact range
                    scales[::97] = np.float32(3e-33)
                    scales[5::211] = np.float32(2e31)
                zps = rng.integers(-5, 6, size=C).astype(np.int32) if with_zp else np.zeros(C, dtype=np.int32)
                shape = (outer, C, inner)
                x32 = _tie_heavy_r6(rng, shape, scales.reshape(1, C, 1), zps.reshape(1, C, 1).astype(np.float32), qmin, qmax)
                xh = _dev(x32).to(tdt)
                x_np = xh.float().cpu().numpy()
                y = torch.full_like(xh, 300.0)
                s_d, z_d = _dev(scales), _dev(zps)
                rc = lib.mctq_fq_per_channel(xh.data_ptr(), y.data_ptr(), outer, C, inner, code, s_d.data_ptr(),
                                             z_d.data_ptr() if with_zp else None, qmin, qmax, _stream())
                assert rc == 0, lib.mctq_last_error()
                seen.add(native.last_launch().split("<")[0])
                want = O.narrow(O.fake_quant_affine(x_np, scales, zps, qmin, qmax, axis=1), dt)
                got = y.float().cpu().numpy()
                assert finite_equal(got, want, x_np), (shape, route, native.last_launch(), first_mismatch(got, want, x_np))
    finally:
        native.set_tuning("shortrows", 1)
    assert {"lastaxis_kernel", "shortrows_kernel"} <= seen, seen


@pytest.mark.parametrize("dt", ["float32", "float16", "bfloat16"])
def test_per_tensor_launches_of_one_round_take_the_paced_kernel_and_agree_with_the_oracle(lib, dt):
    """launch_flat's window (3/4 ... 1 round of resident blocks, 8 per CU) goes through flat_paced_kernel -- the same tile under another
    order of waits -- and nothing outside it does; tuning key "paced": 0 never, 1 the window, 2 whenever a full four-vector tile
    exists.  Sizes on both edges of the window, a partial last tile, n % N trailing elements; a zero point; bit for bit against the
    oracle, and equal to flat_kernel's result."""
    from mct_quantizers_amd.hip import native
    from oracle import mctq_oracle as O
    code, tdt = {"float32": 0, "float16": 1, "bfloat16": 2}[dt], getattr(torch, dt)
    N = 4 if dt == "float32" else 8
    tile = 1024 * N
    rnd = 8 * torch.cuda.get_device_properties(0).multi_processor_count
    lo_edge = -(-3 * rnd // 4)
    sizes = [(lo_edge * tile, True), ((lo_edge - 1) * tile, False), (rnd * tile + N - 1, True), (rnd * tile + N, False),
             ((lo_edge + 7) * tile + tile // 2 + 3, True), ((rnd - 1) * tile + 5 * N + 1, True)]
    scale, zp, qmin, qmax = 0.0371, 3, -128, 127
    rng = np.random.default_rng(77)
    try:
        for n, in_window in sizes:
            x32 = _tie_heavy_r6(rng, (n,), np.float32(scale), np.float32(zp), qmin, qmax)
            xh = _dev(x32).to(tdt)
            x_np = xh.float().cpu().numpy()
            want = O.narrow(O.fake_quant_affine(x_np, np.float32([scale]), np.int32([zp]), qmin, qmax, axis=None), dt)
            got = {}
            for mode in (1, 0, 2):
                native.set_tuning("paced", mode)
                y = torch.full_like(xh, 300.0)
                assert lib.mctq_fq_per_tensor(xh.data_ptr(), y.data_ptr(), n, code, scale, zp, qmin, qmax, _stream()) == 0, lib.mctq_last_error()
                kernel = native.last_launch().split("<")[0]
                assert kernel == ("flat_paced_kernel" if mode == 2 or (mode == 1 and in_window) else "flat_kernel"), (n, mode, native.last_launch())
                got[mode] = y.float().cpu().numpy()
                assert finite_equal(got[mode], want, x_np), (n, dt, mode, native.last_launch(), first_mismatch(got[mode], want, x_np))
            assert bits_equal(got[1], got[0]) and bits_equal(got[2], got[0])
        with pytest.raises(Exception):
            native.set_tuning("paced", 3)
    finally:
        native.set_tuning("paced", 1)


@pytest.mark.parametrize("dt", ["float32", "bfloat16"])
def test_per_channel_launches_of_one_round_take_the_short_row_kernel_with_paced_stores(lib, dt):
    """Symmetric per-channel launches with whole-vector rows that fill 3/4 ... 1 round of shortrows_kernel's tiles run it with paced
    stores; float32 ones are routed there (from rows_kernel / rowsteps_kernel / the gather launch).  Both edges of the window, long
    and short rows, `paced` 0 / 1 / 2, a zero-point table (keeps the old route) -- every result against the oracle bit for bit."""
    from mct_quantizers_amd.hip import native
    from oracle import mctq_oracle as O
    code, tdt = {"float32": 0, "bfloat16": 2}[dt], getattr(torch, dt)
    N = 4 if dt == "float32" else 8
    rnd = 8 * torch.cuda.get_device_properties(0).multi_processor_count
    rows_full = rnd * 1024 * N // 4096                              # rows of 4096 elements that make one round
    shapes = [(rows_full, 4096, True), (rows_full * 3 // 4, 4096, True), (rows_full * 3 // 4 - 8, 4096, False), (rows_full + 8, 4096, False),
              (rows_full * 4, 1024, True), (rows_full * 16, 256, True), (rows_full * 2, 2048, True)]
    rng = np.random.default_rng(99)
    try:
        for C, inner, in_window in shapes:
            scales = rng.uniform(0.01, 0.2, size=C).astype(np.float32)
            x32 = _tie_heavy_r6(rng, (1, C, inner), scales.reshape(1, C, 1), np.float32(0), -128, 127)
            xh = _dev(x32).to(tdt)
            x_np = xh.float().cpu().numpy()
            want = O.narrow(O.fake_quant_affine(x_np, scales, np.zeros(C, dtype=np.int32), -128, 127, axis=1), dt)
            s_d, z_d = _dev(scales), _dev(np.zeros(C, dtype=np.int32))
            names = {}
            for mode, zp in ((1, None), (0, None), (1, z_d)):
                native.set_tuning("paced", mode)
                y = torch.full_like(xh, 300.0)
                rc = lib.mctq_fq_per_channel(xh.data_ptr(), y.data_ptr(), 1, C, inner, code, s_d.data_ptr(), zp.data_ptr() if zp is not None else None,
                                             -128, 127, _stream())
                assert rc == 0, lib.mctq_last_error()
                names[(mode, zp is not None)] = native.last_launch().split("<")[0]
                got = y.float().cpu().numpy()
                assert finite_equal(got, want, x_np), (C, inner, mode, native.last_launch(), first_mismatch(got, want, x_np))
            if dt == "float32":       # routed into shortrows_kernel by the window only (the old routes otherwise, and with zero points)
                assert (names[(1, False)] == "shortrows_kernel") == in_window, (C, inner, names)
                assert names[(0, False)] != "shortrows_kernel" and names[(1, True)] != "shortrows_kernel", (C, inner, names)
    finally:
        native.set_tuning("paced", 1)


def test_fuzz_paced_per_tensor_launch_sizes_vs_oracle(lib):
    """Seeded fuzz of flat_paced_kernel's geometry: element counts anywhere in and around the window (partial last tiles, n % N
    trailing elements, a block more or less than a round), small tensors with the key forced (paced = 2), the three storage types,
    signed / unsigned grids with and without a zero point -- bit for bit against the oracle.  MCTQ_FUZZ_SEED / MCTQ_FUZZ_CASES widen it."""
    import os
    from mct_quantizers_amd.hip import native
    from oracle import mctq_oracle as O
    rng = np.random.default_rng(int(os.environ.get("MCTQ_FUZZ_SEED", "707")))
    rnd = 8 * torch.cuda.get_device_properties(0).multi_processor_count
    seen = set()
    try:
        for case in range(int(os.environ.get("MCTQ_FUZZ_CASES", "24"))):
            dt = ["float32", "float16", "bfloat16"][case % 3]
            code, tdt = {"float32": 0, "float16": 1, "bfloat16": 2}[dt], getattr(torch, dt)
            tile = 1024 * (4 if dt == "float32" else 8)
            forced = case % 4 == 3
            if forced:                                            # any size with a full tile, the key forced
                n = int(rng.integers(tile, 40 * tile)) if rng.random() < 0.5 else int(rng.integers(rnd * tile // 4, rnd * tile // 2))
            else:
                n = int(rng.integers(int(0.70 * rnd) * tile, int(1.04 * rnd) * tile))
            native.set_tuning("paced", 2 if forced else 1)
            qmin, qmax = [(-128, 127), (0, 255), (-8, 7), (0, 15)][int(rng.integers(0, 4))]
            scale, zp = float(rng.uniform(0.01, 0.2)), int(rng.integers(-3, 4)) if rng.random() < 0.5 else 0
            x32 = _tie_heavy_r6(rng, (n,), np.float32(scale), np.float32(zp), qmin, qmax)
            xh = _dev(x32).to(tdt)
            x_np = xh.float().cpu().numpy()
            y = torch.full_like(xh, 300.0)
            assert lib.mctq_fq_per_tensor(xh.data_ptr(), y.data_ptr(), n, code, scale, zp, qmin, qmax, _stream()) == 0, lib.mctq_last_error()
            seen.add(native.last_launch().split("<")[0])
            want = O.narrow(O.fake_quant_affine(x_np, np.float32([scale]), np.int32([zp]), qmin, qmax, axis=None), dt)
            got = y.float().cpu().numpy()
            assert finite_equal(got, want, x_np), (case, n, dt, forced, native.last_launch(), first_mismatch(got, want, x_np))
    finally:
        native.set_tuning("paced", 1)
    assert "flat_paced_kernel" in seen, seen


def test_fuzz_channel_last_and_short_row_launch_geometry_vs_oracle(lib):
    """Seeded fuzz of the round-6 launch geometry through the C ABI: random (outer, C, inner) with inner drawn around the
    lane-vector and tile sizes (1, below a vector, whole vectors, one past / one short of them, rows longer than a tile), C from 1
    to tens of thousands, the three storage types, with and without a zero-point table, every "shortrows" route -- bit for bit
    against the oracle.  MCTQ_FUZZ_SEED / MCTQ_FUZZ_CASES widen it for the soak runs."""
    import os
    from mct_quantizers_amd.hip import native
    from oracle import mctq_oracle as O
    rng = np.random.default_rng(int(os.environ.get("MCTQ_FUZZ_SEED", "606")))
    inners = [1, 1, 1, 2, 3, 4, 5, 7, 8, 9, 12, 15, 16, 17, 20, 24, 31, 32, 33, 40, 63, 64, 65, 100, 127, 128, 200, 252, 255, 256, 257,
              500, 511, 512, 516, 1000, 1020, 1023, 1024, 1025, 1032, 2047, 2048, 2056, 4095, 4096, 4100, 8200, 16385]
    seen = set()
    try:
        for case in range(int(os.environ.get("MCTQ_FUZZ_CASES", "150"))):
            inner = int(rng.choice(inners))
            C = int(rng.choice([1, 2, 3, 5, 7, 8, 13, 16, 31, 64, 100, 200, 255, 256, 768, 1000, 2048, 4096, 4104, 20480, 70001]))
            budget = 1 << int(rng.integers(10, 22))              # elements: a handful of waves ... a few rounds of blocks
            outer = max(1, min(int(rng.integers(1, 9)) if rng.random() < 0.5 else 1 << 20, budget // max(1, C * inner)))
            if outer * C * inner > (1 << 23):
                continue
            dt = ["float32", "float16", "bfloat16"][int(rng.integers(0, 3))]
            code, tdt = {"float32": 0, "float16": 1, "bfloat16": 2}[dt], getattr(torch, dt)
            with_zp = bool(rng.integers(0, 2))
            route = 1 if dt == "float32" and rng.random() < 0.5 else int(rng.integers(0, 3))
            native.set_tuning("shortrows", route)
            qmin, qmax = [(-8, 7), (-128, 127), (0, 255), (0, 3)][int(rng.integers(0, 4))]
            scales = rng.uniform(0.01, 0.2, size=C).astype(np.float32)
            if rng.random() < 0.3:                                # some divisors outside recip_exact's range
                scales[:: int(rng.integers(1, 120))] = np.float32(3e-33)
                scales[int(rng.integers(0, C)):: int(rng.integers(1, 250))] = np.float32(2e31)
            zps = rng.integers(-5, 6, size=C).astype(np.int32) if with_zp else np.zeros(C, dtype=np.int32)
            shape = (outer, C, inner)
            x32 = _tie_heavy_r6(rng, shape, scales.reshape(1, C, 1), zps.reshape(1, C, 1).astype(np.float32), qmin, qmax)
            xh = _dev(x32).to(tdt)
            x_np = xh.float().cpu().numpy()
            y = torch.full_like(xh, 300.0)
            s_d, z_d = _dev(scales), _dev(zps)
            rc = lib.mctq_fq_per_channel(xh.data_ptr(), y.data_ptr(), outer, C, inner, code, s_d.data_ptr(),
                                         z_d.data_ptr() if with_zp else None, qmin, qmax, _stream())
            assert rc == 0, lib.mctq_last_error()
            seen.add(native.last_launch().split("<")[0])
            want = O.narrow(O.fake_quant_affine(x_np, scales, zps, qmin, qmax, axis=1), dt)
            got = y.float().cpu().numpy()
            assert finite_equal(got, want, x_np), (case, shape, dt, with_zp, route, native.last_launch(), first_mismatch(got, want, x_np))
    finally:
        native.set_tuning("shortrows", 1)
    assert {"lastaxis_kernel", "shortrows_kernel", "rows_kernel"} <= seen, seen


def _tie_heavy_r6(rng, shape, s_b, zp_b, qmin, qmax):
    n = int(np.prod(shape))
    with np.errstate(all="ignore"):
        x = (rng.standard_normal(n).astype(np.float32).reshape(shape) * np.minimum(s_b, np.float32(1.0)) * np.float32(0.4 * (qmax - qmin)))
        k = rng.integers(qmin - 2, qmax + 3, size=shape).astype(np.float32) - zp_b
        kind = rng.integers(0, 6, size=shape)
        x = np.where(kind == 0, (k + np.float32(0.5)) * s_b, x)
        x = np.where(kind == 1, k * s_b, x)
    return np.nan_to_num(x.astype(np.float32), nan=0.0, posinf=3e38, neginf=-3e38)


def test_a_tensor_above_the_launch_limit_of_the_short_row_kernels_is_cut_into_row_blocks(lib, monkeypatch):
    """include/mctq_hip.h "Size limit" / ADVICE r05: the LUT, integer-code and export-grid launches of short per-channel rows take
    fewer than 2^32 elements at a time; hip/ops.py cuts larger tensors into row blocks.  With the limit lowered to a few thousand
    elements the block-wise results equal the one-launch results bit for bit."""
    import mct_quantizers_amd as mq
    from mct_quantizers_amd.hip import native, ops
    Q = mq.pytorch_quantizers
    torch.manual_seed(4)
    x = torch.randn(6, 50, 7, 9, device="cuda")
    thr = [0.5 + 0.01 * i for i in range(50)]
    lut = Q.WeightsLUTSymmetricInferableQuantizer(4, [-100.0, -50.0, -10.0, 0.0, 10.0, 50.0, 100.0, 127.0], thr, True, 1, 4)
    s = torch.rand(50, device="cuda") * 0.1 + 0.01
    z = torch.zeros(50, dtype=torch.int32, device="cuda")
    want_lut, want_codes = lut(x), ops.fq_codes(x, s, z, 1, -128, 127)
    want_grid = ops.grid_per_channel(x, -s * 100, s * 100, s, 1)
    xcl = x.contiguous(memory_format=torch.channels_last)
    want_lut_cl = lut(xcl)
    for limit in (2000, 400, 70):
        monkeypatch.setattr(ops, "_SPLIT_ELEMS", limit)
        n0 = native.launch_count()
        assert torch.equal(lut(x), want_lut) and native.launch_count() - n0 > 1
        assert torch.equal(ops.fq_codes(x, s, z, 1, -128, 127), want_codes)
        assert torch.equal(ops.grid_per_channel(x, -s * 100, s * 100, s, 1), want_grid)
        got = lut(xcl)
        assert torch.equal(got, want_lut_cl) and got.is_contiguous()


@pytest.mark.parametrize("dt", ["float32", "bfloat16"])
def test_integer_codes_channel_last_without_a_zero_point_table(lib, dt):
    """mctq_fq_codes_per_channel with zero_points == NULL (all zero) on a channel-last layout: the channel-last kernel's variant
    without a zero-point table -- the ZP form reads the table unconditionally and must not be launched with a NULL one."""
    from oracle import mctq_oracle as O
    rng = np.random.default_rng(12)
    code = {"float32": 0, "bfloat16": 2}[dt]
    rows, C = 777, 4096
    scales = rng.uniform(0.01, 0.2, size=C).astype(np.float32)
    zps = np.zeros(C, dtype=np.int32)
    x32 = _tie_heavy_r6(rng, (rows, C, 1), scales.reshape(1, C, 1), np.float32(0), -128, 127).reshape(rows, C)
    xh = _dev(x32).to(getattr(torch, dt))
    x_np = xh.float().cpu().numpy()
    want = O.fake_quant_affine(x_np, scales, zps, -128, 127, axis=1, return_index=True)[1]
    s_d = _dev(scales)
    for zp_ptr in (None, _dev(zps)):
        codes = torch.full((rows, C), 99, dtype=torch.int8, device="cuda")
        rc = lib.mctq_fq_codes_per_channel(xh.data_ptr(), codes.data_ptr(), rows, C, 1, code, 0, s_d.data_ptr(),
                                           zp_ptr.data_ptr() if zp_ptr is not None else None, -128, 127, _stream())
        assert rc == 0, lib.mctq_last_error()
        torch.cuda.synchronize()
        from mct_quantizers_amd.hip import native
        assert native.last_launch().startswith("lastaxis_kernel<AffineCodesOp"), native.last_launch()
        m = np.isfinite(x_np)
        assert np.array_equal(codes.cpu().numpy().astype(np.int64)[m], want[m])
