"""GPU parity tests of the round-2 additions, all against the oracle / reference fixtures:

float64 tensors, the two bindings of the C ABI (compiled and ctypes), the batched multi-tensor launch, the
tensor-qparams entry point and its fx routing, torch.jit tracing, the literal LUT scan for half inputs, the
remaining config-3 batch sizes, launch-plan invalidation, device checks.
"""
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
# the two bindings of the same C ABI
# ---------------------------------------------------------------------------------------------

def test_compiled_and_ctypes_bindings_agree_with_the_oracle(lib):
    from mct_quantizers_amd.hip import native, ops
    from oracle import mctq_oracle as O
    if os.environ.get("MCTQ_BINDING") == "ctypes" or native.TRACE:
        pytest.skip("the compiled binding is switched off for this run (MCTQ_BINDING=ctypes / MCTQ_ROCTX=1)")
    fast = native.fast()
    assert fast is not None, "the compiled binding must load on the GPU box (python -m mct_quantizers_amd.hip.build)"
    rng = np.random.default_rng(11)
    for shape in ((3, 224, 224), (7,), (5, 1031), (2, 3, 8, 8), (0, 4)):
        x_np = (rng.standard_normal(shape) * 2).astype(np.float32)
        x = _dev(x_np)
        want = O.fake_quant_affine(x_np, [0.0219], [114], 0, 255)
        a = fast.fq_per_tensor(x, 0.0219, 114, 0, 255)
        b = ops._hip_fq_per_tensor(x, 0.0219, 114, 0, 255)
        plan = fast.AffinePlan(0.0219, 114, 0, 255)
        c = plan(x)
        for got in (a, b, c):
            assert got.dtype == x.dtype and got.shape == x.shape and got.stride() == x.stride()
            assert bits_equal(got.cpu().numpy(), want), first_mismatch(got.cpu().numpy(), want, x_np)
    # per channel, every axis, dense permuted storage, Parameter input
    x_np = (rng.standard_normal((6, 5, 40)) * 2).astype(np.float32)
    for axis in (0, 1, 2):
        C = x_np.shape[axis]
        s = rng.uniform(0.01, 0.1, size=C).astype(np.float32)
        z = rng.integers(-3, 4, size=C).astype(np.int32)
        want = O.fake_quant_affine(x_np, s, z, -128, 127, axis=axis)
        sd, zd = _dev(s), _dev(z)
        for x in (_dev(x_np), torch.nn.Parameter(_dev(x_np)), _dev(x_np).permute(2, 0, 1).contiguous().permute(1, 2, 0)):
            a = fast.fq_per_channel(x, sd, zd, axis, -128, 127)
            b = ops._hip_fq_per_channel(x.detach(), sd, zd, axis, -128, 127)
            c = fast.AffinePlan(sd, zd, axis, -128, 127)(x)
            for got in (a, b, c):
                assert got.stride() == x.stride() and not got.requires_grad
                assert bits_equal(got.cpu().numpy(), want), (axis, first_mismatch(got.cpu().numpy(), want, x_np))
    # what the compiled binding declines goes to the general route: NotImplemented, never a wrong answer
    assert fast.fq_per_tensor(torch.randn(4), 0.1, 0, -8, 7) is NotImplemented                      # CPU tensor
    assert fast.fq_per_tensor(torch.randn(8, 8, device="cuda")[:, ::2], 0.1, 0, -8, 7) is NotImplemented   # gaps
    assert fast.fq_per_tensor(torch.zeros(4, device="cuda", dtype=torch.int32), 0.1, 0, -8, 7) is NotImplemented
    assert fast.fq_per_channel(torch.randn(4, 4, device="cuda"), torch.ones(4), None, 0, -8, 7) is NotImplemented  # CPU scales
    y = ops.fq_per_tensor(torch.randn(8, 8, device="cuda")[:, ::2], 0.1, 0, -8, 7)                # ... and still works
    assert y.shape == (8, 4)


def test_ctypes_binding_alone_runs_the_suite_subset(lib, golden_cases, monkeypatch):
    """MCTQ_BINDING=ctypes: same results without the compiled module (fresh ops state)."""
    from mct_quantizers_amd.hip import ops
    monkeypatch.setattr(ops, "_FAST", None)
    monkeypatch.setattr(ops, "_FAST_READY", True)
    cases, arrays = golden_cases
    for c in cases[:40]:
        x_np, want = arrays[c["id"] + "_x"], arrays[c["id"] + "_y"]
        x = _dev(x_np)
        if c["memory_format"] == "channels_last":
            x = x.contiguous(memory_format=torch.channels_last)
        elif c["memory_format"] == "transposed":
            x = x.transpose(0, -1).contiguous().transpose(0, -1)
        q = _make(c["cls"], c["kwargs"])
        assert q.__dict__.get("_plan") in (None, False)
        got = q(x).cpu().numpy()
        assert bits_equal(got, want), f'{c["id"]}: {first_mismatch(got, want, x_np)}'


# ---------------------------------------------------------------------------------------------
# one launch for a list of tensors
# ---------------------------------------------------------------------------------------------

def _batch_cases(rng):
    from oracle import mctq_oracle as O
    items, wants = [], []
    specs = [((64, 4096), 0, torch.float32), ((300, 576), 0, torch.float32), ((7, 33, 5), 1, torch.float32),
             ((16, 8, 3, 3), 0, torch.float32), ((5, 1031), None, torch.float32), ((4, 4096), 1, torch.float32),
             ((3, 10, 10, 6), 3, torch.float32), ((2, 2050), 0, torch.float16), ((9, 257), 1, torch.bfloat16),
             ((128, 1024), 0, torch.bfloat16), ((1,), None, torch.float32), ((6, 37), 0, torch.float64),
             ((11, 23), None, torch.float64)]
    for shape, axis, dt in specs:
        C = 1 if axis is None else shape[axis]
        s = rng.uniform(0.01, 0.1, size=C).astype(np.float32)
        z = rng.integers(-3, 4, size=C).astype(np.int32) if rng.random() < 0.6 else None
        x32 = (rng.standard_normal(shape) * 2).astype(np.float32)
        if dt == torch.float64:
            x = torch.from_numpy(x32.astype(np.float64) * (1 + 1e-9))
            zz = np.zeros(C, np.int32) if z is None else z
            want = O.fake_quant_affine_f64(x.numpy(), s, zz, -128, 127, axis=axis) if axis is not None else \
                O.fake_quant_affine_f64(x.numpy(), s, zz, -128, 127)
        else:
            x = torch.from_numpy(x32).to(dt)
            zz = np.zeros(C, np.int32) if z is None else z
            want = O.narrow(O.fake_quant_affine(x.float().numpy(), s, zz, -128, 127, axis=axis),
                            str(dt).replace("torch.", ""))
        items.append((x.cuda(), _dev(s), None if z is None else _dev(z), axis, -128, 127))
        wants.append(want)
    return items, wants


def test_batched_launch_mixed_shapes_axes_dtypes_against_oracle(lib):
    from mct_quantizers_amd.hip import native, ops
    rng = np.random.default_rng(23)
    items, wants = _batch_cases(rng)
    for route in ("compiled", "ctypes"):
        outs = ops.fq_batched(items) if route == "compiled" else ops._hip_fq_batched(items)
        assert len(outs) == len(items)
        for (x, *_), y, want in zip(items, outs, wants):
            assert y.dtype == x.dtype and y.shape == x.shape and y.stride() == x.stride()
            got = y.cpu().double().numpy() if x.dtype == torch.float64 else y.float().cpu().numpy()
            assert np.array_equal(_bits(got), _bits(want)), (route, tuple(x.shape), x.dtype)
    assert "batched_kernel" in native.last_launch() or "fq64" in native.last_launch() or native.last_launch()
    # more tensors than one launch holds (32), tails, unaligned views (-> single launches inside the call)
    many = []
    for k in range(70):
        n = 1000 + 37 * k
        base = torch.randn(n + 1, device="cuda")
        many.append((base[1:] if k % 5 == 0 else base[:n], _dev(np.float32([0.05 + 0.001 * k])), None, None, -8, 7))
    outs = ops.fq_batched(many)
    for (x, s, _, _, lo, hi), y in zip(many, outs):
        assert torch.equal(y, ops.fq_per_tensor(x.contiguous(), float(s.item()), 0, lo, hi)), x.shape
    assert ops.fq_batched([]) == []
    # a bad descriptor fails before anything is launched
    st = torch.cuda.current_stream().cuda_stream
    arr = (native.FqItem * 2)()
    x = torch.randn(64, device="cuda"); y = torch.full((64,), 7.0, device="cuda"); s = _dev(np.float32([0.1]))
    for it, qmin in zip(arr, (-8, 9)):
        it.x, it.y, it.outer, it.channels, it.inner = x.data_ptr(), y.data_ptr(), 1, 1, 64
        it.scales, it.zero_points, it.quant_min, it.quant_max, it.dtype = s.data_ptr(), None, qmin, 7, native.DT_F32
    assert lib.mctq_fq_batched(arr, 2, st) == native.MCTQ_E_ARG
    torch.cuda.synchronize()
    assert bool((y == 7.0).all())


def test_batched_weight_quantization_of_a_model_is_bit_identical(lib):
    import mct_quantizers_amd as mq
    from mct_quantizers_amd.pytorch.batching import batch_weight_quantization
    Q = mq.pytorch_quantizers
    torch.manual_seed(3)
    layers = []
    for i, (fin, fout) in enumerate(((64, 96), (96, 4096), (4096, 40))):
        lin = torch.nn.Linear(fin, fout)
        thr = [float(v) + 1e-3 for v in lin.weight.detach().abs().amax(dim=1)]
        wq = {"weight": Q.WeightsSymmetricInferableQuantizer(8, thr, True, 0) if i != 1 else
              Q.WeightsUniformInferableQuantizer(4, [-0.3] * fout, [0.2] * fout, True, 0),
              "bias": Q.WeightsSymmetricInferableQuantizer(8, [1.0], False)}
        layers += [mq.PytorchQuantizationWrapper(lin, wq),
                   mq.PytorchActivationQuantizationHolder(Q.ActivationUniformInferableQuantizer(8, [-3.0], [3.0]))]
    lut = mq.PytorchQuantizationWrapper(torch.nn.Linear(40, 8), {"weight": Q.WeightsLUTSymmetricInferableQuantizer(
        3, [-100.0, -50.0, -10.0, 0.0, 10.0, 50.0, 100.0, 127.0], [1.0], False)})      # stays on its own quantizer
    model = torch.nn.Sequential(*layers, lut).cuda()
    x = torch.randn(5, 64, device="cuda")
    want = model(x)
    want_w = [m.layer.weight.clone() for m in model if isinstance(m, mq.PytorchQuantizationWrapper)]
    handle = batch_weight_quantization(model)
    assert handle.quantize_now() == 6
    got = model(x)
    assert torch.equal(got, want)
    for m, w in zip([m for m in model if isinstance(m, mq.PytorchQuantizationWrapper)], want_w):
        assert torch.equal(m.layer.weight, w) and "_prequantized" not in m.__dict__
    with torch.no_grad():
        model[0].weight.mul_(0.5)                           # weights are re-quantized on EVERY forward
    after = model(x)
    handle.remove()
    assert torch.equal(model(x), after) and not torch.equal(after, want)


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


def test_traced_reference_weights_quantizers_are_rerouted_and_bit_exact(lib):
    """tests/golden/ref_traced_weight_quantizers.pth: fx trace of the REFERENCE's per-tensor / per-channel weights
    quantizers (tensor-qparams nodes).  Every fake_quantize node must end up on the mctq_amd ops and reproduce the
    reference's outputs."""
    from mct_quantizers_amd import compat
    gm = compat.load_reference_model(os.path.join(GOLDEN, "ref_traced_weight_quantizers.pth"), map_location="cuda")
    targets = [str(n.target) for n in gm.graph.nodes if n.op == "call_function"]
    assert sum("fq_per_tensor_tqp" in t for t in targets) == 2 and sum("fq_per_channel" in t for t in targets) == 1
    assert not any("fake_quantize" in t for t in targets), targets
    io = np.load(os.path.join(GOLDEN, "ref_traced_weight_quantizers_io.npz"))
    o1, o3, o2 = gm(_dev(io["w"]), _dev(io["v"]))
    for got, key in ((o1, "o1"), (o3, "o3"), (o2, "o2")):
        assert got.is_cuda and bits_equal(got.cpu().numpy(), io[key]), key
    # the traced reference WRAPPER: weights were folded at trace time, the activation node is rerouted
    gw = compat.load_reference_model(os.path.join(GOLDEN, "ref_traced_wrapper.pth"), map_location="cuda").cuda()
    assert any("mctq_amd" in str(n.target) for n in gw.graph.nodes if n.op == "call_function")
    iow = np.load(os.path.join(GOLDEN, "ref_traced_wrapper_io.npz"))
    assert bits_equal(gw.l1.layer.weight.cpu().numpy(), iow["w1"]) and bits_equal(gw.l2.layer.weight.cpu().numpy(), iow["w2"])
    y = gw(_dev(iow["x"])).detach().cpu().numpy()
    assert np.allclose(y, iow["y"], rtol=0, atol=1e-4)              # two float32 GEMMs on another device


# ---------------------------------------------------------------------------------------------
# torch.jit tracing without enable_custom_impl (TorchScript / fakely-quant ONNX export)
# ---------------------------------------------------------------------------------------------

def test_jit_trace_records_aten_nodes_not_an_uninitialised_buffer(lib):
    import mct_quantizers_amd as mq
    Q = mq.pytorch_quantizers
    x = torch.randn(4, 16, device="cuda")
    holder = mq.PytorchActivationQuantizationHolder(Q.ActivationUniformInferableQuantizer(8, [-2.5], [3.1]))
    lin = torch.nn.Linear(16, 8)
    wrapper = mq.PytorchQuantizationWrapper(lin, {"weight": Q.WeightsSymmetricInferableQuantizer(8, [0.3] * 8, True, 0),
                                                  "bias": Q.WeightsSymmetricInferableQuantizer(8, [0.5], False)}).cuda()
    lutq = Q.ActivationLutPOTInferableQuantizer(2, [-100.0, 0.0, 60.0, 127.0], [4.0], True)
    wl = Q.WeightsLUTSymmetricInferableQuantizer(3, [-100.0, -50.0, -10.0, 0.0, 10.0, 50.0, 100.0, 127.0], [1.0, 2.0, 0.5, 1.5], True, 0, 2)
    for mod, inp in ((holder, x), (wrapper, x), (mq.PytorchActivationQuantizationHolder(lutq), x),
                     (mq.PytorchActivationQuantizationHolder(Q.ActivationPOTInferableQuantizer(4, [2.0], True)), x)):
        eager = mod(inp)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            traced = torch.jit.trace(mod, inp, check_trace=False)
        kinds = [n.kind() for n in traced.graph.nodes()]
        if mod is not wrapper:
            assert any("fake_quantize" in k or "argmin" in k for k in kinds), kinds
        assert not any(k == "aten::empty_like" for k in kinds), kinds
        y1, y2 = traced(inp), traced(inp * 0.5)
        assert torch.equal(y1, eager) and torch.equal(y2, mod(inp * 0.5))
    w = torch.randn(4, 33, device="cuda")
    f = lambda t: wl(t)   # noqa: E731
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        tr = torch.jit.trace(f, w.clone(), check_trace=False)
    assert torch.equal(tr(w.clone()), wl(w.clone()))
    assert torch.equal(mq.PytorchActivationQuantizationHolder(lutq)(x), lutq(x))      # eager path untouched afterwards


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


def test_holder_fast_call_keeps_module_semantics(lib):
    import mct_quantizers_amd as mq
    Q = mq.pytorch_quantizers
    h = mq.PytorchActivationQuantizationHolder(Q.ActivationUniformInferableQuantizer(8, [-2.5], [3.1]))
    x = torch.randn(2, 8, device="cuda")
    want = h.forward(x)
    assert torch.equal(h(x), want)
    seen = []
    hook = h.register_forward_hook(lambda m, i, o: seen.append(1))
    assert torch.equal(h(x), want) and seen == [1]
    hook.remove()
    pre = h.register_forward_pre_hook(lambda m, i: (i[0] * 0,))
    assert float(h(x).abs().sum()) == float(h.forward(x * 0).abs().sum())
    pre.remove()
    b = mq.PytorchFLNActivationQuantizationHolder(Q.ActivationPOTInferableQuantizer(8, [2.0], True), quantization_bypass=True)
    assert b(x) is x


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


def test_first_call_of_a_process_inside_graph_capture(lib, tmp_path):
    """The per-tensor kernel is launched through hipModuleLaunchKernel with its hipFunction_t resolved on first use
    (and the compiled binding is imported lazily): both must be legal when the very first quantizer call of a process
    happens under hipGraph stream capture."""
    import subprocess
    import sys
    from conftest import REPO
    code = f"""
import sys, torch, logging
sys.path.insert(0, {REPO!r})
logging.getLogger("mct_quantizers_amd").setLevel(logging.ERROR)
import mct_quantizers_amd as mq
Q = mq.pytorch_quantizers
x = torch.randn(2, 3, 32, 32, device="cuda")
w = torch.randn(64, 4096, device="cuda")
qa = Q.ActivationUniformInferableQuantizer(8, [-2.5], [3.1])
qw = Q.WeightsSymmetricInferableQuantizer(8, [1.0 + 0.01 * i for i in range(64)], True, 0)
scale_t = torch.tensor([0.02], device="cuda")
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    ya = qa(x)
    yw = qw(w)
    from mct_quantizers_amd.hip import ops
    yb = ops.fq_batched([qw.batch_item(w), (x, scale_t, None, None, 0, 255)])
x.copy_(torch.randn_like(x)); w.copy_(torch.randn_like(w))
g.replay(); torch.cuda.synchronize()
assert torch.equal(ya, torch.fake_quantize_per_tensor_affine(x, qa.scale, qa.zero_point, 0, 255))
assert torch.equal(yw, torch.fake_quantize_per_channel_affine(w, qw.scales, qw.zero_points, 0, -128, 127))
assert torch.equal(yb[0], yw) and torch.equal(yb[1], torch.fake_quantize_per_tensor_affine(x, 0.02, 0, 0, 255))
print("capture-ok")
"""
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "capture-ok" in r.stdout, r.stderr[-2000:]


def test_batched_weight_quantization_with_persistent_buffers(lib):
    """reuse_buffers=True: pre-packed BatchPlan + persistent outputs.  Same values as per-layer quantization on every
    forward, in-place weight updates followed, sub-module calls past the hook fall back to the quantizer, refresh()
    picks up changed quantizer parameters, remove() restores the reference behaviour."""
    import mct_quantizers_amd as mq
    from mct_quantizers_amd.pytorch.batching import batch_weight_quantization
    Q = mq.pytorch_quantizers
    torch.manual_seed(5)

    def build():
        torch.manual_seed(5)
        mods = []
        for fin, fout in ((64, 96), (96, 4096), (4096, 48)):
            lin = torch.nn.Linear(fin, fout)
            thr = [float(v) + 1e-3 for v in lin.weight.detach().abs().amax(dim=1)]
            mods += [mq.PytorchQuantizationWrapper(lin, {"weight": Q.WeightsSymmetricInferableQuantizer(8, thr, True, 0),
                                                         "bias": Q.WeightsUniformInferableQuantizer(8, [-0.5], [0.5], False)}),
                     mq.PytorchActivationQuantizationHolder(Q.ActivationUniformInferableQuantizer(8, [-3.0], [3.0]))]
        return torch.nn.Sequential(*mods).cuda()

    ref, model = build(), build()
    x = torch.randn(7, 64, device="cuda")
    handle = batch_weight_quantization(model, reuse_buffers=True)
    y1 = model(x)
    from mct_quantizers_amd.hip import native
    if native.fast() is None:                                         # MCTQ_BINDING=ctypes or MCTQ_ROCTX=1
        assert handle._plan is None and torch.equal(y1, ref(x))       # no BatchPlan without the compiled binding:
        pytest.skip("compiled binding switched off: per-forward batching stands in (checked), the plan itself needs it")
    assert handle._plan is not None and torch.equal(y1, ref(x))
    w_obj = model[0].layer.weight
    for _ in range(3):
        with torch.no_grad():
            for m, r in zip(model, ref):
                if isinstance(m, mq.PytorchQuantizationWrapper):
                    m.weight.mul_(0.9); r.weight.mul_(0.9)           # in-place update: pointers unchanged, values new
        assert torch.equal(model(x), ref(x))
        assert model[0].layer.weight is w_obj                        # the persistent tensor, rewritten in place
    # a wrapper called directly (past the model's pre-hook) must not serve the previous generation's tensor
    with torch.no_grad():
        model[0].weight.mul_(0.5); ref[0].weight.mul_(0.5)
    assert torch.equal(model[0](x), ref[0](x))
    assert torch.equal(model(x), ref(x))
    # changed quantizer parameters: refresh()
    q, qr = model[2].weights_quantizers["weight"], ref[2].weights_quantizers["weight"]
    q.scales = q.scales * 2.0; qr.scales = qr.scales * 2.0
    handle.refresh()
    assert torch.equal(model(x), ref(x))
    handle.remove()
    assert torch.equal(model(x), ref(x)) and model[0].layer.weight is not w_obj
    assert all("_prequantized_plan" not in m.__dict__ for m in model)


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


def test_capture_forward_replays_the_model_and_follows_weight_updates(lib):
    import mct_quantizers_amd as mq
    Q = mq.pytorch_quantizers

    def build():
        torch.manual_seed(9)
        mods = []
        cin = 16
        for cout, k in ((32, 3), (32, 1), (16, 3)):
            conv = torch.nn.Conv2d(cin, cout, k, padding=k // 2)
            thr = [float(v) + 1e-6 for v in conv.weight.detach().abs().amax(dim=(1, 2, 3))]
            mods += [mq.PytorchQuantizationWrapper(conv, {"weight": Q.WeightsSymmetricInferableQuantizer(8, thr, True, 0)}),
                     mq.PytorchActivationQuantizationHolder(Q.ActivationUniformInferableQuantizer(8, [-3.0], [3.0]))]
            cin = cout
        mods.append(mq.PytorchActivationQuantizationHolder(Q.ActivationLutPOTInferableQuantizer(
            2, [-100.0, 0.0, 60.0, 127.0], [4.0], True)))
        return torch.nn.Sequential(*mods).cuda()

    ref, model = build(), build()
    x = torch.randn(2, 16, 12, 12, device="cuda")
    for batch_weights in (True, False):
        cap = mq.capture_forward(model, x, batch_weights=batch_weights)
        with torch.no_grad():
            assert torch.equal(cap(x), ref(x))
            x2 = torch.randn_like(x)
            assert torch.equal(cap(x2), ref(x2))
            for m, r in zip(model, ref):                             # in-place weight update: seen by the next replay
                if isinstance(m, mq.PytorchQuantizationWrapper):
                    m.weight.mul_(0.8); r.weight.mul_(0.8)
            assert torch.equal(cap(x2), ref(x2))
        with pytest.raises(ValueError):
            cap(torch.randn(3, 16, 12, 12, device="cuda"))
        cap.release()
        with torch.no_grad():
            assert torch.equal(model(x), ref(x))


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
# torch.compile (aot_eager): the traced graph calls the mctq_amd:: library ops, which launch the HIP kernels
# ---------------------------------------------------------------------------------------------

def test_torch_compile_runs_the_hip_ops_on_the_gpu(lib):
    import mct_quantizers_amd as mq
    from mct_quantizers_amd.hip import native
    from oracle import mctq_oracle as O
    Q = mq.pytorch_quantizers
    torch.manual_seed(3)
    lin = torch.nn.Linear(64, 24).cuda()
    thr = [0.5 + 0.05 * i for i in range(24)]
    lut = [-128.0, -64.0, -20.0, -5.0, 0.0, 5.0, 20.0, 64.0]
    m = torch.nn.Sequential(
        mq.PytorchActivationQuantizationHolder(Q.ActivationUniformInferableQuantizer(8, [-2.5], [3.1])),
        mq.PytorchQuantizationWrapper(lin, {"weight": Q.WeightsSymmetricInferableQuantizer(8, thr, True, 0)}),
        mq.PytorchActivationQuantizationHolder(Q.ActivationLutPOTInferableQuantizer(3, lut, [4.0], True)),
    ).cuda()
    x = torch.randn(32, 64, device="cuda")
    want = m(x)
    torch._dynamo.reset()
    cm = torch.compile(m, backend="aot_eager", fullgraph=True)         # one graph: no break at any quantizer
    got = cm(x)
    assert torch.equal(got, want)
    ex = torch._dynamo.explain(m)(x)
    targets = [str(n.target) for g in ex.graphs for n in g.graph.nodes if n.op == "call_function"]
    assert ex.graph_break_count == 0 and {"mctq_amd.fq_per_tensor", "mctq_amd.fq_per_channel", "mctq_amd.lut_per_tensor"} <= set(targets)
    assert "kernel" in native.last_launch()                              # a HIP kernel of this library ran last
    # each quantizer's own output inside the compiled graph is the oracle's
    xq = torch.compile(m[0], backend="aot_eager")(x)
    s, z, qmin, qmax, _, _ = O.activation_uniform_params(8, [-2.5], [3.1])
    assert bits_equal(xq.cpu().numpy(), O.fake_quant_affine(x.cpu().numpy(), np.float32(s), z, qmin, qmax))


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
    assert kinds <= {"LutTableOp", "LutStepsOp", "LutCellsOp"} and kinds, kinds


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


def test_bench_multi_process_leg_runs_over_rccl_on_one_gpu(lib):
    """bench.py launched as the driver launches it for N > 1 (torch.distributed.run, one process per GPU), here with one
    rank on the one GPU of the box: RCCL initialises, the barrier / max-over-ranks / sharded config-5 leg with its
    all_gather_into_tensor execute, and the JSON line carries the keys the scaling run reads."""
    import json
    import subprocess
    import sys
    from conftest import REPO
    import socket
    with socket.socket() as sk:                           # a port that is free right now (the rendezvous store binds it)
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(REPO, "bench.py"), "--gpus", "1", "--gather", "--steps", "20", "--warmup", "5",
           "--no-cpu", "--prewarm-seconds", "0.2", "--evidence-launches", "0"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=REPO)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-1000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["scaling"] == "weak" and d["value"] > 1e11
    assert d["config"]["control_plane"] == "nccl", d["config"]
    leg = d["sharded_cfg5"]
    for key in ("compute_ms", "compute_elems_per_s", "allgather_ms", "allgather_recv_bytes_per_rank", "compute_plus_allgather_elems_per_s", "gathered_rows_match_local"):
        assert key in leg, leg
    assert leg["gathered_rows_match_local"] is True and leg["rows_per_rank"] == 8192


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


@pytest.mark.parametrize("world", [2, 8, 3])
def test_dim0_shards_reassemble_to_the_reference_digest_of_config_5(lib, world):
    """The multi-GPU partition of SURVEY 8(e) at FULL size on one GPU: each rank's row block of the 8192 x 8192 tensor,
    quantized with that rank's slice of the thresholds (sharded.shard_kwargs / row_block), concatenates to the tensor
    whose SHA-256 the reference produced (tests/golden/full_sha.json) -- even and ragged (3 ranks) splits."""
    import mct_quantizers_amd as mq
    from mct_quantizers_amd import sharded, workloads
    rec = load_json("full_sha.json")["configs"]["cfg5"]
    x_np = workloads.make_input("cfg5")
    wl = workloads.make_workload("cfg5", x_np)
    rows = x_np.shape[0]
    h = hashlib.sha256()
    covered = 0
    for rank in range(world):
        start, stop = sharded.row_block(rows, world, rank)
        q = getattr(mq.pytorch_quantizers, wl.quantizer)(**sharded.shard_kwargs(wl.kwargs, rows, world, rank))
        y = q(_dev(x_np[start:stop]))
        h.update(np.ascontiguousarray(y.cpu().numpy()).tobytes())
        covered += stop - start
    assert covered == rows and h.hexdigest() == rec["y_sha256"]


def test_integration_md_stub_runs_as_written(lib):
    """The reference-side ctypes stub printed in INTEGRATION.md (blocks 1-3) is executed verbatim against the built
    library and its per-channel replacement compared with the ATen operator it replaces."""
    import re
    from conftest import REPO
    from mct_quantizers_amd.hip import native
    text = open(os.path.join(REPO, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, flags=re.S)
    assert len(blocks) >= 3
    src = "\n".join(blocks[:3]).replace('ctypes.CDLL("libmctq_hip.so")', f"ctypes.CDLL({native.lib_path()!r})")
    ns = {}
    exec(compile(src, "INTEGRATION.md", "exec"), ns)
    x = torch.randn(6, 40, 7, device="cuda")
    s = torch.rand(40, device="cuda") * 0.1 + 0.01
    z = torch.randint(-5, 6, (40,), dtype=torch.int32, device="cuda")
    got = ns["fake_quantize_per_channel_hip"](x, s, z, 1, -128, 127)
    assert torch.equal(got, torch.fake_quantize_per_channel_affine(x, s, z, 1, -128, 127))
    assert ctypes_sizeof(ns["FqItem"]) == 72
    # the table-form hook of the same document (block with `class BatchedWeights`), run as written on three quantizers
    import mct_quantizers_amd as mq
    Q = mq.pytorch_quantizers
    hook_src = next(b for b in blocks if "class BatchedWeights" in b)
    exec(compile(hook_src, "INTEGRATION.md", "exec"), ns)
    torch.manual_seed(1)
    ws = [torch.randn(40, 96, device="cuda"), torch.randn(8, 3, 5, 5, device="cuda"), torch.randn(1000, device="cuda")]
    qs = [Q.WeightsSymmetricInferableQuantizer(8, [1.0 + 0.1 * i for i in range(40)], True, 0),
          Q.WeightsUniformInferableQuantizer(4, [-1.0] * 3, [2.0] * 3, True, 1),
          Q.WeightsSymmetricInferableQuantizer(8, [3.0], False)]
    outs = ns["BatchedWeights"](list(zip(ws, qs)))()
    torch.cuda.synchronize()
    for w, q, y in zip(ws, qs, outs):
        assert torch.equal(y, q(w.clone())), type(q).__name__


def ctypes_sizeof(t):
    import ctypes
    return ctypes.sizeof(t)
