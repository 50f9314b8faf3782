"""The ONNX-export branch: ``enable_custom_impl()`` + ``torch.jit`` tracing.

Fixtures (tests/golden/export_cases.*) were produced by the reference on CPU with tools/gen_golden.py
--export-only: outputs of its export-time arithmetic on tie / clip-edge / non-finite inputs, and the
``mct_quantizers::*`` nodes (attributes + constant inputs) its symbolic functions emit.
Bar: bit-exact outputs (NaN matches NaN; signed zeros must agree), identical nodes.
"""
import json
import os
import warnings

import numpy as np
import pytest
import torch

from conftest import GOLDEN


def _load():
    with open(os.path.join(GOLDEN, "export_cases.json")) as f:
        meta = json.load(f)
    return meta, np.load(os.path.join(GOLDEN, "export_cases.npz"))


def _same(got, want):
    got = np.ascontiguousarray(got, dtype=np.float32)
    want = np.ascontiguousarray(want, dtype=np.float32)
    if got.shape != want.shape:
        return False, f"shape {got.shape} vs {want.shape}"
    ok = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))
    if ok.all():
        return True, ""
    i = tuple(np.argwhere(~ok)[0])
    return False, f"{int((~ok).sum())} mismatches, first at {i}: got={got[i]!r} want={want[i]!r}"


def _traced_call(q, x):
    """q(x) evaluated while torch.jit is tracing (what torch.onnx.export does to the model)."""
    box = {}

    def fn(t):
        box["y"] = q(t)
        return box["y"]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        torch.jit.trace(fn, x, check_trace=False)
    return box["y"].detach()


def _make(case):
    import mct_quantizers_amd as mq
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        q = getattr(mq.pytorch_quantizers, case["cls"])(**case["kwargs"])
    q.enable_custom_impl()
    return q


def _onnx_nodes(module, example):
    utils = pytest.importorskip("torch.onnx._internal.torchscript_exporter.utils")
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
            attrs[a] = n.t(a).tolist() if kind == "t" else getattr(n, "is_" if kind == "is" else kind)(a)
        out.append([n.kind(), attrs, len(list(n.inputs()))])
    return [n for n in out if n[0].startswith("mct_quantizers::") or n[0] == "onnx::Constant"]


class _WeightHolder(torch.nn.Module):
    def __init__(self, quant, shape):
        super().__init__()
        self.q = quant
        self.w = torch.nn.Parameter(torch.zeros(shape))

    def forward(self, t):
        return t + self.q(self.w).sum()


# ------------------------------------------------------------------------------------------------
# CPU
# ------------------------------------------------------------------------------------------------

def test_oracle_matches_reference_export_arithmetic():
    from oracle import oracle_export_call
    meta, arrays = _load()
    assert len(meta["cases"]) >= 50
    for c in meta["cases"]:
        ok, why = _same(oracle_export_call(c["cls"], c["kwargs"], arrays[c["id"] + "_x"]), arrays[c["id"] + "_y"])
        assert ok, f'{c["id"]} {c["name"]}: {why}'


def test_traced_quantizers_on_cpu_tensors_match_reference():
    meta, arrays = _load()
    for c in meta["cases"]:
        q = _make(c)
        y = _traced_call(q, torch.from_numpy(arrays[c["id"] + "_x"].copy()))
        ok, why = _same(y.numpy(), arrays[c["id"] + "_y"])
        assert ok, f'{c["id"]} {c["name"]}: {why}'


def test_exported_nodes_match_reference():
    import mct_quantizers_amd as mq
    meta, _ = _load()
    want_by_id = {n["id"]: n["nodes"] for n in meta["onnx_nodes"]}
    for c in meta["cases"]:
        q = _make(c)
        if c["cls"].startswith("Weights"):
            mod, ex = _WeightHolder(q, c["shape"]), torch.zeros(3)
        else:
            mod, ex = mq.PytorchActivationQuantizationHolder(q), torch.zeros(c["shape"])
        got = _onnx_nodes(mod, ex)
        assert json.loads(json.dumps(got)) == want_by_id[c["id"]], f'{c["id"]} {c["name"]}'


def test_export_branch_needs_both_the_flag_and_tracing():
    import mct_quantizers_amd as mq
    Q = mq.pytorch_quantizers
    x = torch.tensor([0.05, 0.299, -0.31, 1.7, -3.0])
    q = Q.ActivationSymmetricInferableQuantizer(num_bits=3, threshold=[1.3], signed=True)
    plain = q(x)
    assert torch.equal(_traced_call(q, x), plain)                 # tracing alone: still the fake-quant kernel
    q.enable_custom_impl()
    assert torch.equal(q(x), plain)                               # flag alone: same
    # both: the export arithmetic (true division) -- equal here up to rounding, and on the same grid
    y = _traced_call(q, x)
    assert torch.allclose(y, plain, atol=1e-6)


def test_export_branch_honours_the_reuse_cache():
    import mct_quantizers_amd as mq
    q = mq.pytorch_quantizers.WeightsSymmetricInferableQuantizer(num_bits=4, threshold=[1.0, 2.0], per_channel=True,
                                                                  channel_axis=0)
    q.enable_custom_impl()
    q.enable_reuse_quantizer()
    w = torch.randn(2, 5)
    first = _traced_call(q, w)
    assert q.resue_outputs is not None and not q.quantizer_first_run
    assert torch.equal(q(torch.randn(2, 5)), first)              # cached object served afterwards


def test_public_export_names_exist():
    from mct_quantizers_amd.pytorch.quantizers import onnx_export as E
    for name in ("WeightsSymmetricF", "WeightsPOTF", "WeightsUniformF", "WeightsLUTSymmetricF", "WeightsLUTPOTF",
                 "ActivationSymF", "ActivationPOTF", "ActivationUniformF", "BaseQuantizerAutogradFunction",
                 "quantize_sym_weights_torch", "quantize_uniform_weights_torch", "quantize_sym_activations_torch",
                 "quantize_uniform_activations_torch"):
        assert hasattr(E, name), name
    with pytest.raises(NotImplementedError):
        E.WeightsSymmetricF.backward(None, None)


# ------------------------------------------------------------------------------------------------
# GPU: the same branch on device tensors runs the gfx950 grid kernel (mctq_grid_per_*_f32)
# ------------------------------------------------------------------------------------------------

@pytest.mark.gpu
def test_traced_quantizers_on_gpu_match_reference_goldens():
    from mct_quantizers_amd.hip import native
    native.load()
    meta, arrays = _load()
    for c in meta["cases"]:
        q = _make(c)
        y = _traced_call(q, torch.from_numpy(arrays[c["id"] + "_x"].copy()).cuda())
        assert y.is_cuda
        ok, why = _same(y.cpu().numpy(), arrays[c["id"] + "_y"])
        assert ok, f'{c["id"]} {c["name"]}: {why}'


@pytest.mark.gpu
@pytest.mark.parametrize("shape,axis", [((4096, 1024), 0), ((64, 300, 7), 1), ((33, 5, 257), 2), ((1000003,), None)])
def test_grid_kernel_matches_oracle_on_large_inputs(shape, axis):
    from oracle import mctq_oracle as O
    from mct_quantizers_amd.hip import ops
    rng = np.random.default_rng(int(np.prod(shape)) + 7 * (axis or 0))
    x = (rng.standard_normal(shape) * 3).astype(np.float32)
    for shifted in (False, True):
        if axis is None:
            lo, hi, step = -2.5035293, 3.0964706, 0.021960784
            y = ops.grid_per_tensor(torch.from_numpy(x).cuda(), lo, hi, step, shifted)
            want = O.export_grid(x, lo, hi, step, None, shifted)
        else:
            C = shape[axis]
            lo = -rng.uniform(0.1, 4, C).astype(np.float32)
            hi = rng.uniform(0.1, 4, C).astype(np.float32)
            step = ((hi - lo) / np.float32(255)).astype(np.float32)
            y = ops.grid_per_channel(torch.from_numpy(x).cuda(), torch.from_numpy(lo), torch.from_numpy(hi),
                                     torch.from_numpy(step), axis, shifted)
            want = O.export_grid(x, lo, hi, step, axis, shifted)
        ok, why = _same(y.cpu().numpy(), want)
        assert ok, f"{shape} axis={axis} shifted={shifted}: {why}"


@pytest.mark.gpu
def test_grid_kernel_all_float32_inputs_against_torch_cpu_chain():
    """Every float32 bit pattern through the grid kernel vs the reference's torch op chain on the CPU."""
    from mct_quantizers_amd.hip import ops
    lo, hi, step = np.float32(-1.3), np.float32(1.3 - 1.3 / 128), np.float32(1.3 / 128)
    lo_t, hi_t, st_t = (torch.tensor(float(v), dtype=torch.float32) for v in (lo, hi, step))
    chunk = 1 << 27
    torch.set_num_threads(min(32, torch.get_num_threads()))       # more threads than that only thrash
    for start in range(0, 1 << 32, chunk):
        bits = torch.arange(start, start + chunk, dtype=torch.int64, device="cuda").to(torch.int32)   # wraps: all patterns
        xd = bits.view(torch.float32)
        del bits
        y = ops.grid_per_tensor(xd, float(lo), float(hi), float(step), False).cpu()
        x = xd.cpu()
        del xd
        c = torch.where(x < lo_t, lo_t, x)
        want = torch.round(torch.where(x > hi_t, hi_t, c) / st_t) * st_t
        same = (y.view(torch.int32) == want.view(torch.int32)) | (torch.isnan(y) & torch.isnan(want))
        assert bool(same.all()), f"chunk at {start:#x}: {int((~same).sum())} mismatches"


@pytest.mark.gpu
def test_grid_entry_points_validate_arguments():
    from mct_quantizers_amd.hip import native
    lib = native.load()
    x = torch.zeros(8, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    assert lib.mctq_grid_per_tensor_f32(x.data_ptr(), x.data_ptr(), -1, 0.0, 1.0, 0.1, 0, s) == native.MCTQ_E_ARG
    assert lib.mctq_grid_per_tensor_f32(x.data_ptr(), x.data_ptr(), 8, 0.0, 1.0, 0.1, 2, s) == native.MCTQ_E_ARG
    assert lib.mctq_grid_per_channel_f32(x.data_ptr(), x.data_ptr(), 1, 8, 1, None, None, None, 0, s) == native.MCTQ_E_ARG
    assert lib.mctq_grid_per_tensor_f32(None, None, 0, 0.0, 1.0, 0.1, 0, s) == 0          # empty is fine
    # other storage types (export of a half-precision model, once per export): the reference's op chain on the device,
    # the same values as that chain on the CPU copy
    from mct_quantizers_amd.hip import ops
    xh = (torch.randn(5, 33, device="cuda") * 0.7).half()
    got = ops.grid_per_tensor(xh, -1.0, 0.875, 0.125)
    assert got.is_cuda and torch.equal(got.cpu(), ops.grid_per_tensor(xh.cpu(), -1.0, 0.875, 0.125))
    lo, hi, st = torch.tensor([-1.0] * 5), torch.tensor([0.875] * 5), torch.tensor([0.125] * 5)
    got = ops.grid_per_channel(xh, lo, hi, st, 0)
    assert got.is_cuda and torch.equal(got.cpu(), ops.grid_per_channel(xh.cpu(), lo, hi, st, 0))
