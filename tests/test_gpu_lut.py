"""GPU parity of the LUT quantizers beyond test_gpu_parity.py: the literal scan on half inputs, the sorted threshold list
(wide codebooks) against the literal scan for ALL 2^32 inputs and against the oracle, the quantizer classes' choice of
kernel, fuzz over shapes / axes / layouts / dtypes / codebook widths, the clip bounds of half-precision activations,
attribute assignment, integer tensors.  (float64 lists: test_lut_f64_steps.py.)"""
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
# literal LUT scan for half-precision storage (no decision table)
# ---------------------------------------------------------------------------------------------

def test_literal_scan_takes_half_inputs_and_step_rounding(lib):
    from mct_quantizers_amd.hip import native, ops
    from oracle import mctq_oracle as O
    rng = np.random.default_rng(41)
    lut = [-128.0, -96.0, -64.0, -40.0, -24.0, -12.0, -5.0, 0.0, 5.0, 12.0, 24.0, 40.0, 64.0, 96.0, 120.0, 127.0]
    lut_d = _dev(np.float32(lut))
    x32 = (rng.standard_normal((5, 700)) * 1.4).astype(np.float32)
    for dt, name, code in ((torch.float16, "float16", native.DT_F16), (torch.bfloat16, "bfloat16", native.DT_BF16)):
        x = torch.from_numpy(x32).to(dt)
        xw = x.float().numpy()
        # activation flavour: Python-float threshold, per-step roundings in the tensor's type
        div = float(torch.tensor([2.0 + 1e-8], dtype=torch.float64).to(dt).item())
        want = O.lut_quantize(xw, lut, 2.0, True, 8, 1e-8, step_dtype=name)
        got = ops._hip_lut_per_tensor(x.cuda(), lut_d, div, 2.0, 128.0, -128.0, 127.0, None, code)
        assert got.dtype == torch.float32 and bits_equal(got.cpu().numpy(), want), (name, first_mismatch(got.cpu().numpy(), want, xw))
        # weights flavour: float32 tensor threshold -> promoted chain, per channel, NO widening pass
        thr = rng.uniform(0.5, 3.0, size=5).astype(np.float32)
        want = O.lut_quantize(xw, lut, thr, True, 8, 1e-8, per_channel=True, channel_axis=0)
        got = ops._hip_lut_per_channel(x.cuda(), lut_d, _dev(thr), 1e-8, 0, 128.0, -128.0, 127.0, None)
        assert bits_equal(got.cpu().numpy(), want), name
        assert "LutOp" in native.last_launch() and "in2B" in native.last_launch(), native.last_launch()


# ---------------------------------------------------------------------------------------------
# wide integer codebooks (lut_values_bitwidth > 10): sorted threshold list, binary search in LDS
# ---------------------------------------------------------------------------------------------

def _wide_codebooks():
    rng = np.random.default_rng(77)
    return {
        "s12_l16": ([float(v) for v in rng.choice(np.arange(-2048, 2048), 16, replace=False)], True, 12),
        "s12_l3dup": ([3.0, 3.0, -8.0], True, 12),
        "u12_l64": ([float(v) for v in rng.choice(np.arange(0, 4097), 64, replace=False)], False, 12),
        "s16_l256": ([float(v) for v in rng.choice(np.arange(-32768, 32768), 256, replace=False)], True, 16),
        "s11_l5": ([-1024.0, 1023.0, 0.0, 1.0, -1.0], True, 11),
        "s16_l1024": ([float(v) for v in rng.choice(np.arange(-32768, 32768), 1024, replace=False)], True, 16),
    }


def _domain(signed, B):
    mult = float(2 ** (B - int(signed)))
    return (mult, float(-2 ** (B - 1)), float(2 ** (B - 1) - 1)) if signed else (mult, 0.0, float(2 ** B - 1))


@pytest.mark.parametrize("name", list(_wide_codebooks()))
def test_threshold_list_equals_literal_scan_for_every_float(lib, name):
    """All 2^32 float32 inputs: the threshold-list kernel == the literal first-minimum scan kernel."""
    from mct_quantizers_amd.hip import native
    lut, signed, B = _wide_codebooks()[name]
    mult, cmin, cmax = _domain(signed, B)
    st = native.build_lut_steps(lut, mult, cmin, cmax)
    assert st is not None
    lut_d, st_d = _dev(np.asarray(lut, dtype=np.float32)), _dev(st)
    chunk = 1 << 28
    y_lit = torch.empty(chunk, dtype=torch.float32, device="cuda")
    y_st = torch.empty(chunk, dtype=torch.float32, device="cuda")
    for c in range(16):
        bits = torch.arange(c * chunk - (1 << 31), (c + 1) * chunk - (1 << 31), dtype=torch.int64, device="cuda")
        x = bits.to(torch.int32).view(torch.float32)
        del bits
        # thr_div = thr_mul = 1: t = clamp(x * mult) sweeps every float of the clip range
        assert lib.mctq_lut_per_tensor_f32(x.data_ptr(), y_lit.data_ptr(), chunk, 1.0, 1.0, lut_d.data_ptr(),
                                           len(lut), mult, cmin, cmax, _stream()) == 0
        assert lib.mctq_luts_per_tensor(x.data_ptr(), y_st.data_ptr(), chunk, native.DT_F32, 0, 1.0, 1.0,
                                        st_d.data_ptr(), st_d.numel(), mult, cmin, cmax, _stream()) == 0, lib.mctq_last_error()
        if not torch.equal(y_lit.view(torch.int32), y_st.view(torch.int32)):
            i = int(torch.nonzero(y_lit.view(torch.int32) != y_st.view(torch.int32))[0])
            raise AssertionError(f"chunk {c}: x={x[i].item()!r} literal={y_lit[i].item()!r} steps={y_st[i].item()!r}")
        del x
    assert ("LutCellsOp" if len(set(lut)) > 64 else "LutStepsOp") in native.last_launch(), native.last_launch()


@pytest.mark.parametrize("name", ["s12_l16", "u12_l64", "s16_l256"])
@pytest.mark.parametrize("outer,C,inner", [(1, 3, 1), (4, 6, 5), (2, 6, 1024), (3, 5, 1028), (1, 16, 11008), (1, 3000, 3),
                                           (41, 64, 1), (3, 4096, 1)])
def test_threshold_list_per_channel_vs_oracle(lib, name, outer, C, inner):
    from mct_quantizers_amd.hip import native
    from oracle import mctq_oracle as O
    lut, signed, B = _wide_codebooks()[name]
    mult, cmin, cmax = _domain(signed, B)
    rng = np.random.default_rng(C * 7 + inner)
    thr = rng.uniform(0.05, 4.0, size=C).astype(np.float32)
    shape = (outer, C, inner)
    tb = thr.reshape(1, C, 1)
    x_np = rng.standard_normal(shape).astype(np.float32) * tb * np.float32(0.7)
    # a third of the elements at midpoints between adjacent centres (ties / hand-overs)
    srt = np.sort(np.unique(np.float32(lut)))
    mids = (srt[:-1] + srt[1:]) * np.float32(0.5)
    pick = mids[rng.integers(0, mids.size, size=shape)] / np.float32(mult) * tb
    x_np = np.where(rng.integers(0, 3, size=shape) == 0, pick, x_np).astype(np.float32)
    x_np.reshape(-1)[:3] = np.float32([0.0, -0.0, 1e30])[: min(3, x_np.size)]
    st_d = _dev(native.build_lut_steps(lut, mult, cmin, cmax))
    want = O.lut_quantize(x_np, lut, thr, signed, B, 1e-8, per_channel=True, channel_axis=1)
    for dt, code in ((torch.float32, native.DT_F32), (torch.float16, native.DT_F16), (torch.bfloat16, native.DT_BF16)):
        x = _dev(x_np).to(dt)
        t_d = _dev(thr)
        y = torch.empty(shape, dtype=torch.float32, device="cuda")
        rc = lib.mctq_luts_per_channel(x.data_ptr(), y.data_ptr(), outer, C, inner, code, t_d.data_ptr(), 1e-8,
                                       st_d.data_ptr(), st_d.numel(), mult, cmin, cmax, _stream())
        assert rc == 0, lib.mctq_last_error()
        if dt is not torch.float32:
            xw = x.float().cpu().numpy()
            want_h = O.lut_quantize(xw, lut, thr, signed, B, 1e-8, per_channel=True, channel_axis=1)
            assert bits_equal(y.cpu().numpy(), want_h), (dt, first_mismatch(y.cpu().numpy(), want_h, xw))
        else:
            assert bits_equal(y.cpu().numpy(), want), first_mismatch(y.cpu().numpy(), want, x_np)


def test_wide_codebook_quantizer_classes_take_the_threshold_list(lib):
    """lut_values_bitwidth = 12 / 16 through the reference's classes: bit-equal to the oracle, launched as LutStepsOp;
    a non-integer codebook (operator layer only) still runs the literal scan."""
    import mct_quantizers_amd as mq
    from mct_quantizers_amd.hip import native
    from oracle import mctq_oracle as O
    Q = mq.pytorch_quantizers
    rng = np.random.default_rng(5)
    for name in ("s12_l16", "s16_l256"):
        lut, signed, B = _wide_codebooks()[name]
        nb = int(np.log2(len(lut)))
        thr = rng.uniform(0.5, 3.0, size=8).astype(np.float32)
        w_np = (rng.standard_normal((8, 33, 3, 3)) * 1.1).astype(np.float32)
        q = Q.WeightsLUTSymmetricInferableQuantizer(nb, lut, [float(t) for t in thr], True, 0, 4, lut_values_bitwidth=B)
        assert q._lut_table_torch is None and q._lut_steps_torch is not None
        got = q(_dev(w_np))
        assert "LutStepsOp" in native.last_launch() or "LutCellsOp" in native.last_launch(), native.last_launch()
        want = O.lut_quantize(w_np, lut, thr, True, B, 1e-8, per_channel=True, channel_axis=0)
        assert bits_equal(got.cpu().numpy(), want)
        q1 = Q.WeightsLUTPOTInferableQuantizer(nb, lut, [2.0], False, lut_values_bitwidth=B)
        got = q1(_dev(w_np))
        assert "LutStepsOp" in native.last_launch() or "LutCellsOp" in native.last_launch(), native.last_launch()
        assert bits_equal(got.cpu().numpy(), O.lut_quantize(w_np, lut, np.float32([2.0]), True, B, 1e-8))
        # activation quantizer, float32 and half inputs (per-step half roundings)
        qa = Q.ActivationLutPOTInferableQuantizer(nb, lut, [4.0], True, lut_values_bitwidth=B)
        x_np = (rng.standard_normal((4, 3, 17, 19)) * 2.0).astype(np.float32)
        got = qa(_dev(x_np))
        assert "LutStepsOp" in native.last_launch() or "LutCellsOp" in native.last_launch(), native.last_launch()
        assert bits_equal(got.cpu().numpy(), O.lut_quantize(x_np, lut, 4.0, True, B, 1e-8))
        for dt, dname in ((torch.float16, "float16"), (torch.bfloat16, "bfloat16")):
            xh = torch.from_numpy(x_np).to(dt)
            got = qa(xh.cuda())
            assert "LutStepsOp" in native.last_launch() or "LutCellsOp" in native.last_launch(), native.last_launch()
            want = O.lut_quantize(xh.float().numpy(), lut, 4.0, True, B, 1e-8, step_dtype=dname)
            assert finite_equal(got.float().cpu().numpy(), want, xh.float().numpy()), dname
    # the classes only accept integer codebooks (base_lut_symmetric_inferable_quantizer.py:66); a non-integer one handed
    # to the operator layer has neither table nor threshold list and runs the literal scan
    from mct_quantizers_amd.hip import ops
    assert ops.make_lut_steps(np.float32([-100.5, 3.25, 7.0, 900.0]), 2048.0, -2048.0, 2047.0, "cuda") is None
    w_np = (rng.standard_normal((64, 65)) * 1.1).astype(np.float32)
    got = ops.lut_per_tensor(_dev(w_np), _dev(np.float32([-100.5, 3.25, 7.0, 900.0])), 1.5, 1.5, 2048.0, -2048.0, 2047.0)
    assert "LutOp" in native.last_launch(), native.last_launch()
    assert bits_equal(got.cpu().numpy(), O.lut_quantize(w_np, [-100.5, 3.25, 7.0, 900.0], np.float32([1.5]), True, 12, 0.0))


# ---------------------------------------------------------------------------------------------
# fuzz of the LUT quantizer classes: table / threshold-list / float64 kernels vs the torch op chain on CPU and the oracle
# ---------------------------------------------------------------------------------------------

def test_fuzz_lut_quantizers_shapes_axes_layouts_dtypes_and_codebook_widths(lib):
    """Seeded fuzz over ranks, shapes, channel axes (negative ones too), permuted storage, gapped views, storage types,
    codebook sizes and lut_values_bitwidth 4..16: the HIP result equals the op chain the reference runs on the CPU copy of
    the same tensor (this package's CPU route = torch ops in the reference's order), and, for float32, the oracle."""
    import os
    import mct_quantizers_amd as mq
    from mct_quantizers_amd.hip import native
    from oracle import mctq_oracle as O
    Q = mq.pytorch_quantizers
    rng = np.random.default_rng(int(os.environ.get("MCTQ_FUZZ_SEED", "77")))
    seen = set()
    for case in range(int(os.environ.get("MCTQ_FUZZ_CASES", "120"))):
        rank = int(rng.integers(1, 5))
        shape = [int(rng.choice([1, 2, 3, 4, 5, 7, 8, 16, 33, 64])) for _ in range(rank)]
        if rng.random() < 0.3:
            shape[int(rng.integers(0, rank))] = int(rng.choice([257, 1024, 1030, 4096]))
        if int(np.prod(shape, dtype=np.int64)) > (1 << 22):
            continue
        B = int(rng.choice([4, 8, 8, 10, 12, 16]))
        nb = int(rng.integers(1, min(B, 6) + 1))
        dt = [torch.float32, torch.float32, torch.float16, torch.bfloat16, torch.float64][int(rng.integers(0, 5))]
        kind = int(rng.integers(0, 3))                       # 0 weights per channel, 1 weights per tensor, 2 activation
        signed = True if kind < 2 else bool(rng.integers(0, 2))
        lo, hi = (-2 ** (B - 1), 2 ** (B - 1)) if signed else (0, 2 ** B + 1)
        lut = [float(v) for v in rng.choice(np.arange(lo, hi), int(rng.integers(1, 2 ** nb + 1)), replace=False)]
        axis = int(rng.integers(0, rank))
        x = (torch.from_numpy(rng.standard_normal(shape).astype(np.float32)) * 1.5).to(dt)
        perm = list(rng.permutation(rank))
        x = x.permute(perm).contiguous().permute(list(np.argsort(perm)))
        if rng.random() < 0.2 and shape[0] > 1:
            x = x[::2]
        C = x.shape[axis]
        if kind == 0:
            thr = [float(v) for v in rng.uniform(0.3, 3.0, size=C)]
            ax = axis - rank if rng.random() < 0.3 else axis
            q = Q.WeightsLUTSymmetricInferableQuantizer(nb, lut, thr, True, ax, rank, lut_values_bitwidth=B)
        elif kind == 1:
            thr = [float(2.0 ** rng.integers(-2, 3))]
            q = Q.WeightsLUTPOTInferableQuantizer(nb, lut, thr, False, lut_values_bitwidth=B)
        else:
            thr = [float(2.0 ** rng.integers(-2, 3))]
            q = Q.ActivationLutPOTInferableQuantizer(nb, lut, thr, signed, lut_values_bitwidth=B)
        # CPU copy of the tensor and of the parameters: torch ops in the reference's order (ops._cpu_lut_*)
        from mct_quantizers_amd.hip import ops
        from mct_quantizers_amd.pytorch.quantizers.lut import lut_domain
        mult, cmin, cmax = lut_domain(B, signed)
        lut_t = torch.tensor(lut, dtype=torch.float32)
        if kind == 0:
            want = ops._cpu_lut_per_channel(x.clone(), lut_t, torch.tensor(thr, dtype=torch.float32), 1e-8, axis, mult, cmin, cmax)
        elif kind == 1:
            want = ops._cpu_lut_per_tensor(x.clone(), lut_t, q._thr_div0, q._thr_mul0, mult, cmin, cmax, 0)
        else:
            step = {torch.float16: native.DT_F16, torch.bfloat16: native.DT_BF16}.get(dt, 0)
            try:
                want = ops._cpu_lut_per_tensor(x.clone(), lut_t, float(thr[0]) + 1e-8, q._thr_mul0, mult, cmin, cmax, step or -1)
            except RuntimeError as e:                         # float16 tensor, clip bound 65535: torch refuses; so do we
                with pytest.raises(RuntimeError) as e2:
                    q(x.cuda())
                assert str(e2.value) == str(e)
                continue
        got = q(x.cuda())
        seen.add(native.last_launch().split("<")[1].split(",")[0] if "<" in native.last_launch() else native.last_launch())
        info = (case, tuple(x.shape), x.stride(), axis, dt, kind, B, len(lut))
        assert got.dtype == want.dtype and got.shape == want.shape and got.is_contiguous() and want.is_contiguous(), info
        if dt in (torch.float16, torch.bfloat16) and kind == 2:
            ok = finite_equal(got.float().cpu().numpy(), want.float().numpy(), x.float().numpy())
        else:
            ok = bits_equal(got.float().cpu().numpy(), want.float().numpy())
        assert ok, (info, first_mismatch(got.float().cpu().numpy(), want.float().numpy(), x.float().numpy()))
        if dt is torch.float32:
            thr_o = np.float32(thr) if kind < 2 else thr[0]
            w = O.lut_quantize(x.numpy(), lut, thr_o, signed, B, 1e-8, per_channel=(kind == 0),
                               channel_axis=(axis if kind == 0 else None))
            assert bits_equal(got.cpu().numpy(), w), info
    assert {"LutTableOp", "LutStepsOp"} <= seen, seen


def test_half_activation_lut_clip_bounds_follow_the_tensor_type(lib):
    """float16 / bfloat16 activations with lut_values_bitwidth 9..16 (reference fixtures, cases_half_bounds): the clip
    range is the one torch.clip uses on that tensor type (511 -> 512 in bfloat16 ...), served by a decision table /
    threshold list built for THAT range; the float16 configuration torch refuses raises the same RuntimeError."""
    import json
    import warnings
    import mct_quantizers_amd as mq
    from mct_quantizers_amd.hip import native
    Q = mq.pytorch_quantizers
    with open(os.path.join(GOLDEN, "cases_half_bounds.json")) as f:
        cases = json.load(f)["cases"]
    arrays = np.load(os.path.join(GOLDEN, "cases_half_bounds.npz"))
    kinds = set()
    for c in cases:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            q = Q.ActivationLutPOTInferableQuantizer(**c["kwargs"])
        x32 = arrays[c["id"] + "_x"]
        x = _dev(x32).to(getattr(torch, c["in_dtype"]))
        if "error" in c:
            with pytest.raises(RuntimeError) as e:
                q(x)
            assert str(e.value) == c["error"]
            continue
        y = q(x)
        kinds.add(native.last_launch().split("<")[1].split(",")[0])
        want = arrays[c["id"] + "_y"]
        assert y.is_cuda and str(y.dtype) == "torch." + c["out_dtype"], c["id"]
        assert finite_equal(y.float().cpu().numpy(), want, x32), (c["id"], first_mismatch(y.float().cpu().numpy(), want, x32))
        # the float32 twin of the same quantizer keeps the float32 range
        y32 = q(_dev(x32))
        from oracle import mctq_oracle as O
        kw = c["kwargs"]
        assert bits_equal(y32.cpu().numpy(), O.lut_quantize(x32, kw["lut_values"], kw["threshold"][0], kw["signed"],
                                                            kw["lut_values_bitwidth"], 1e-8))
    assert kinds <= {"LutTableOp", "LutCompactOp", "LutStepsOp", "LutCellsOp"} and kinds, kinds


def test_lut_quantizers_follow_attribute_assignment(lib):
    """The reference reads threshold / eps / lut_values (and the weights classes' private tensors) on every call; here
    tables and pre-packed launches are derived from them, so assigning to one must re-derive that state."""
    import mct_quantizers_amd as mq
    from oracle import mctq_oracle as O
    Q = mq.pytorch_quantizers
    rng = np.random.default_rng(11)
    lut = [-128.0, -64.0, -20.0, -5.0, 0.0, 5.0, 20.0, 64.0]
    x_np = (rng.standard_normal((3, 257)) * 3).astype(np.float32)
    x = _dev(x_np)
    q = Q.ActivationLutPOTInferableQuantizer(3, lut, [4.0], True)
    assert bits_equal(q(x).cpu().numpy(), O.lut_quantize(x_np, lut, 4.0, True, 8, 1e-8))
    q.threshold = 8.0
    assert bits_equal(q(x).cpu().numpy(), O.lut_quantize(x_np, lut, 8.0, True, 8, 1e-8))
    q.eps = 0.25
    assert bits_equal(q(x).cpu().numpy(), O.lut_quantize(x_np, lut, 8.0, True, 8, 0.25))
    q.lut_values = torch.tensor([-100.0, 0.0, 100.0], device="cuda")
    assert bits_equal(q(x).cpu().numpy(), O.lut_quantize(x_np, [-100.0, 0.0, 100.0], 8.0, True, 8, 0.25))
    q.lut_values_bitwidth = 10
    q.lut_values = torch.tensor([-400.0, 3.0, 300.0], device="cuda")
    assert bits_equal(q(x).cpu().numpy(), O.lut_quantize(x_np, [-400.0, 3.0, 300.0], 8.0, True, 10, 0.25))
    w = Q.WeightsLUTSymmetricInferableQuantizer(3, lut, [1.0, 2.0, 0.5], True, 0, 2)
    assert bits_equal(w(_dev(x_np)).cpu().numpy(), O.lut_quantize(x_np, lut, np.float32([1.0, 2.0, 0.5]), True, 8, 1e-8, per_channel=True, channel_axis=0))
    w._threshold_torch = torch.tensor([3.0, 0.25, 1.5], device="cuda")
    w._lut_values_torch = torch.tensor([-7.0, 1.0, 90.0], device="cuda")
    assert bits_equal(w(_dev(x_np)).cpu().numpy(), O.lut_quantize(x_np, [-7.0, 1.0, 90.0], np.float32([3.0, 0.25, 1.5]), True, 8, 1e-8, per_channel=True, channel_axis=0))
    wt = Q.WeightsLUTPOTInferableQuantizer(3, lut, [2.0], False)
    wt(_dev(x_np))
    wt.eps = 0.5                                            # enters the per-tensor divisor
    assert bits_equal(wt(_dev(x_np)).cpu().numpy(), O.lut_quantize(x_np, lut, np.float32([2.0]), True, 8, 0.5))


def test_lut_quantizers_take_integer_tensors_like_the_reference_chain(lib):
    """The reference's LUT chain starts with a true division, which promotes integer tensors to float32; the affine
    operators (ATen) refuse them -- both behaviours are kept."""
    import mct_quantizers_amd as mq
    from mct_quantizers_amd.hip import ops
    Q = mq.pytorch_quantizers
    lut = [-128.0, -64.0, -20.0, -5.0, 0.0, 5.0, 20.0, 64.0]
    lut_t = torch.tensor(lut)
    qa = Q.ActivationLutPOTInferableQuantizer(3, lut, [4.0], True)
    qw = Q.WeightsLUTSymmetricInferableQuantizer(3, lut, [1.0, 2.0, 0.5], True, 1, 2)
    for dt in (torch.int32, torch.int64, torch.uint8, torch.bool):
        x = torch.randint(0, 2 if dt is torch.bool else 7, (5, 3)).to(dt)
        want = ops._cpu_lut_per_tensor(x, lut_t, 4.0 + 1e-8, 4.0, 128.0, -128.0, 127.0, -1)
        got = qa(x.cuda())
        assert got.dtype == torch.float32 and torch.equal(got.cpu(), want), dt
        want = ops._cpu_lut_per_channel(x, lut_t, torch.tensor([1.0, 2.0, 0.5]), 1e-8, 1, 128.0, -128.0, 127.0)
        got = qw(x.cuda())
        assert torch.equal(got.cpu(), want), dt
    with pytest.raises(NotImplementedError):
        Q.ActivationUniformInferableQuantizer(8, [-2.5], [3.1])(torch.ones(3, dtype=torch.int32, device="cuda"))
