"""GPU: the drop-in boundary -- the two bindings of the C ABI (compiled, ctypes) agree with the oracle, fx traces of the
reference's quantizers are re-routed to the HIP ops, torch.jit traces record ATen's nodes, torch.compile runs the library
ops as one graph, and the ctypes stub printed in INTEGRATION.md runs as written."""
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
