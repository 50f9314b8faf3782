"""Host-side behaviour of the drop-in API on CPU tensors (no GPU): constructors, registry, reuse
cache, containers, fx trace + pickle round trips.  Modelled on the reference's own unit tests
(tests/pytorch_tests/**), with the golden fixtures supplying the expected numbers."""
import io
import pickle
import warnings

import numpy as np
import pytest
import torch

import mct_quantizers_amd as mq
from conftest import bits_equal, finite_equal, first_mismatch, load_json
from mct_quantizers_amd import (PytorchActivationQuantizationHolder, PytorchFLNActivationQuantizationHolder,
                                PytorchPreservingActivationQuantizationHolder, PytorchQuantizationWrapper,
                                QuantizationMethod, QuantizationTarget, get_inferable_quantizer_class)

Q = mq.pytorch_quantizers


def test_cpu_tensors_take_the_aten_route_and_match_goldens(golden_cases):
    """BASELINE config 1 plumbing: a CPU tensor gets exactly what the reference gives it."""
    cases, arrays = golden_cases
    for c in cases:
        x = torch.from_numpy(arrays[c["id"] + "_x"])
        if c["memory_format"] == "channels_last":
            x = x.contiguous(memory_format=torch.channels_last)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            q = getattr(Q, c["cls"])(**c["kwargs"])
        got = q(x).numpy()
        want = arrays[c["id"] + "_y"]
        assert bits_equal(got, want), f'{c["id"]} {c["cls"]}: {first_mismatch(got, want)}'


def test_half_precision_cpu_tensors_match_goldens(half_cases):
    cases, arrays = half_cases
    for c in cases:
        x32 = arrays[c["id"] + "_x"]
        x = torch.from_numpy(x32).to(getattr(torch, c["in_dtype"]))
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            q = getattr(Q, c["cls"])(**c["kwargs"])
        y = q(x)
        assert str(y.dtype) == "torch." + c["out_dtype"], c["id"]
        assert finite_equal(y.float().numpy(), arrays[c["id"] + "_y"], x32), f'{c["id"]} {c["cls"]} {c["in_dtype"]}'


def test_constructor_attributes_match_reference():
    for rec in load_json("ctor.json")["ctor"]:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            q = getattr(Q, rec["cls"])(**rec["kwargs"])
        for name, want in rec["attrs"].items():
            got = getattr(q, name)
            if isinstance(want, dict):
                vals = got.detach().cpu().numpy() if isinstance(got, torch.Tensor) else np.asarray(got)
                assert str(vals.dtype) in want["dtype"], (rec["cls"], name, vals.dtype, want["dtype"])
                assert np.array_equal(vals.reshape(-1).astype(np.float64), np.asarray(want["values"])), (rec["cls"], name)
            else:
                assert got == want, (rec["cls"], name, got, want)


def test_registry_finds_exactly_one_class_per_target_and_method():
    base = Q.BasePyTorchInferableQuantizer
    expect = {
        (QuantizationTarget.Weights, QuantizationMethod.SYMMETRIC): Q.WeightsSymmetricInferableQuantizer,
        (QuantizationTarget.Weights, QuantizationMethod.POWER_OF_TWO): Q.WeightsPOTInferableQuantizer,
        (QuantizationTarget.Weights, QuantizationMethod.UNIFORM): Q.WeightsUniformInferableQuantizer,
        (QuantizationTarget.Weights, QuantizationMethod.LUT_SYM_QUANTIZER): Q.WeightsLUTSymmetricInferableQuantizer,
        (QuantizationTarget.Weights, QuantizationMethod.LUT_POT_QUANTIZER): Q.WeightsLUTPOTInferableQuantizer,
        (QuantizationTarget.Activation, QuantizationMethod.SYMMETRIC): Q.ActivationSymmetricInferableQuantizer,
        (QuantizationTarget.Activation, QuantizationMethod.POWER_OF_TWO): Q.ActivationPOTInferableQuantizer,
        (QuantizationTarget.Activation, QuantizationMethod.UNIFORM): Q.ActivationUniformInferableQuantizer,
        (QuantizationTarget.Activation, QuantizationMethod.LUT_POT_QUANTIZER): Q.ActivationLutPOTInferableQuantizer,
    }
    for (target, method), cls in expect.items():
        assert get_inferable_quantizer_class(target, method, base) is cls
    with pytest.raises(Exception):
        get_inferable_quantizer_class(QuantizationTarget.Activation, QuantizationMethod.LUT_SYM_QUANTIZER, base)


def test_reuse_cache_returns_the_same_object():
    for q in (Q.WeightsSymmetricInferableQuantizer(8, [1.0], False),
              Q.WeightsPOTInferableQuantizer(8, [2.0], False),
              Q.WeightsUniformInferableQuantizer(8, [-1.0], [1.0], False),
              Q.WeightsLUTSymmetricInferableQuantizer(2, [-25.0, 25.0], [1.0], False),
              Q.WeightsLUTPOTInferableQuantizer(2, [-25.0, 25.0], [1.0], False)):
        assert q.resue_outputs is None and q.quantizer_first_run and not q.enable_reuse
        a = q(torch.randn(4, 4))
        assert q.resue_outputs is None                      # reuse is off by default
        q.enable_reuse_quantizer()
        b = q(torch.randn(4, 4))
        c = q(torch.randn(4, 4))
        assert c is b and q.resue_outputs is b and not q.quantizer_first_run
        q.disable_reuse_quantizer()
        assert q(torch.randn(4, 4)) is not b
        assert a is not b


def test_weights_quantizers_flip_requires_grad_and_activations_do_not():
    w = torch.nn.Parameter(torch.randn(3, 8))
    Q.WeightsSymmetricInferableQuantizer(8, [1.0, 1.0, 1.0], True, 0)(w)
    assert not w.requires_grad
    x = torch.randn(3, 8, requires_grad=True)
    y = Q.ActivationUniformInferableQuantizer(8, [-1.0], [1.0])(x)
    assert x.requires_grad and not y.requires_grad


class _ZeroWeightsQuantizer(mq.BaseInferableQuantizer):
    def __call__(self, inputs):
        return inputs * 0


def test_wrapper_named_weights():
    conv = torch.nn.Conv2d(3, 20, 3)
    wrapper = PytorchQuantizationWrapper(conv, {"weight": _ZeroWeightsQuantizer()})
    assert wrapper.is_weights_quantization and wrapper.num_weights_quantizers == 1
    (name, weight, quantizer), = wrapper.get_weights_vars()
    assert name == "weight" and isinstance(weight, torch.nn.Parameter) and isinstance(quantizer, _ZeroWeightsQuantizer)
    assert isinstance(wrapper.layer, torch.nn.Conv2d)
    out = wrapper(torch.randn(1, 3, 8, 8))
    bias = conv.bias.detach().reshape(1, -1, 1, 1)
    assert torch.allclose(out, bias.expand_as(out))          # all weights quantized to zero
    assert torch.all(wrapper.get_quantized_weights()["weight"] == 0)


def test_wrapper_positional_weights_and_list_inputs():
    # torch.sub(const, x)
    wrapper = PytorchQuantizationWrapper(torch.sub, {0: _ZeroWeightsQuantizer()}, {0: torch.ones(3)})
    x = torch.tensor([1.0, 2.0, 3.0])
    assert torch.equal(wrapper(x), -x)
    assert "positional_weight_0" in dict(wrapper.named_parameters())
    # torch.cat([c0, x, c2], dim=1)
    wrapper = PytorchQuantizationWrapper(torch.cat, {0: _ZeroWeightsQuantizer(), 2: _ZeroWeightsQuantizer()},
                                         {0: torch.ones(2, 1), 2: torch.ones(2, 2)}, op_call_kwargs={"dim": 1},
                                         is_inputs_as_list=True)
    out = wrapper(torch.full((2, 3), 7.0))
    assert out.shape == (2, 6) and torch.all(out[:, 0] == 0) and torch.all(out[:, 1:4] == 7) and torch.all(out[:, 4:] == 0)


def test_wrapper_argument_checks_raise():
    with pytest.raises(Exception, match="should be a torch.Tensor"):
        PytorchQuantizationWrapper(torch.sub, {0: _ZeroWeightsQuantizer()}, {0: [1.0]})
    with pytest.raises(Exception, match="keys should be all strings"):
        PytorchQuantizationWrapper(torch.nn.Linear(2, 2), {0: _ZeroWeightsQuantizer()})
    with pytest.raises(Exception, match="Mismatch"):
        PytorchQuantizationWrapper(torch.sub, {1: _ZeroWeightsQuantizer()}, {0: torch.ones(1)})


def test_wrapper_with_a_real_quantizer_requantizes_every_forward():
    lin = torch.nn.Linear(16, 4, bias=False)
    thr = [float(v) for v in lin.weight.detach().abs().max(dim=1).values]
    q = Q.WeightsSymmetricInferableQuantizer(4, thr, True, 0)
    wrapper = PytorchQuantizationWrapper(lin, {"weight": q})
    x = torch.randn(2, 16)
    y1 = wrapper(x)
    wq = wrapper.layer.weight
    assert torch.unique(wq[0]).numel() <= 16
    assert torch.equal(y1, x @ wq.t())
    with torch.no_grad():
        wrapper.weight.mul_(0.5)                              # float weight changes -> next forward re-quantizes
    y2 = wrapper(x)
    assert not torch.equal(y1, y2)


@pytest.mark.parametrize("holder_cls", [PytorchActivationQuantizationHolder, PytorchFLNActivationQuantizationHolder,
                                        PytorchPreservingActivationQuantizationHolder])
def test_holders_quantize_and_bypass(holder_cls):
    q = Q.ActivationPOTInferableQuantizer(num_bits=3, threshold=[4.0], signed=True)
    holder = holder_cls(q)
    x = torch.randn(1, 3, 16, 16) * 3
    y = holder(x)
    assert torch.unique(y).numel() <= 8 and y.min() >= -4 and y.max() <= 3
    assert torch.equal(y, q(x))
    if holder_cls is not PytorchActivationQuantizationHolder:
        assert holder.quantization_bypass is False
        bypass = holder_cls(q, quantization_bypass=True)
        assert bypass(x) is x


@pytest.mark.parametrize("make", [
    lambda: Q.ActivationUniformInferableQuantizer(3, [-2.0], [2.0]),
    lambda: Q.ActivationSymmetricInferableQuantizer(8, [3.0], True),
    lambda: Q.ActivationLutPOTInferableQuantizer(3, [-25.0, 25.0, 100.0], [4.0], True),
])
def test_fx_trace_then_pickle_round_trip(make):
    """reference tests/pytorch_tests/test_activation_quantizer_holder.py:67-90."""
    holder = PytorchActivationQuantizationHolder(make())
    x = torch.randn(2, 3, 8, 8)
    want = holder(x)
    traced = torch.fx.symbolic_trace(holder)
    targets = [n.target for n in traced.graph.nodes if n.op == "call_function"]
    assert any("mctq_amd" in str(t) for t in targets), targets     # one opaque op, routed per device at run time
    assert torch.equal(traced(x), want)
    buf = io.BytesIO()
    torch.save(traced, buf)
    buf.seek(0)
    loaded = torch.load(buf, weights_only=False)
    assert torch.equal(loaded(x), want)


class _Net(torch.nn.Module):
    def __init__(self):
        super().__init__()
        conv = torch.nn.Conv2d(3, 4, 3)
        thr = [float(v) for v in conv.weight.detach().abs().amax(dim=(1, 2, 3))]
        self.conv = PytorchQuantizationWrapper(conv, {"weight": Q.WeightsSymmetricInferableQuantizer(8, thr, True, 0)})
        self.act = PytorchActivationQuantizationHolder(Q.ActivationUniformInferableQuantizer(8, [-1.0], [3.0]))
        lin = torch.nn.Linear(4, 2)
        self.lin = PytorchQuantizationWrapper(lin, {"weight": Q.WeightsLUTSymmetricInferableQuantizer(
            3, [-100.0, -50.0, -10.0, 0.0, 10.0, 50.0, 100.0, 127.0], [1.0], False)})

    def forward(self, x):
        return self.lin(self.act(self.conv(x)).mean(dim=(2, 3)))


def test_whole_module_pickle_round_trip(tmp_path):
    """reference tests/pytorch_tests/test_pytorch_load_model.py: torch.save(module) / load / same output."""
    net = _Net()
    x = torch.randn(2, 3, 10, 10)
    want = net(x)
    path = tmp_path / "model.pth"
    torch.save(net, path)
    loaded = mq.pytorch_load_quantized_model(path)
    assert torch.equal(loaded(x), want)
    pickle.loads(pickle.dumps(net.act.activation_holder_quantizer))        # quantizers hold no ctypes handles


def test_utils_surface():
    from mct_quantizers_amd.pytorch import quantizer_utils as U
    assert U.get_working_device().type in ("cpu", "cuda")
    t = U.to_torch_tensor(np.asarray([1.0, 2.0]))
    assert t.dtype == torch.float32
    assert U.to_torch_tensor(3).dtype == torch.int32 and U.to_torch_tensor(3.0).dtype == torch.float32
    assert isinstance(U.to_torch_tensor([np.zeros(2), 1.0]), list)
    with pytest.raises(Exception):
        U.to_torch_tensor("nope")
    lo, hi = U.fix_range_to_include_zero(torch.tensor([-4.0, 3.0, -7.0]), torch.tensor([4.0, 10.0, -1.0]), 7)
    assert np.isclose(lo[0].item(), -4.03149606299213, atol=1e-6) and np.isclose(hi[0].item(), 3.96850393700787, atol=1e-6)
    assert lo[1].item() == 0.0 and hi[1].item() == 10.0 and lo[2].item() == -7.0 and hi[2].item() == 0.0
    # lut_quantizer keeps the reference signature
    lut = torch.tensor([-25.0, 25.0])
    y = U.lut_quantizer(torch.tensor([[-1.0, 0.3]]), lut, True, torch.tensor([2.0]), 8, 1e-8)
    assert torch.allclose(y, torch.tensor([[-25.0 / 128 * 2, 25.0 / 128 * 2]]))


def test_integer_codes_dequantize_to_the_fake_quant_output_cpu():
    """Extension API: (codes - zero_point) * scale reproduces the fake-quantized tensor exactly."""
    x = torch.randn(6, 40) * 2
    for q in (Q.WeightsSymmetricInferableQuantizer(8, [0.5, 1.0, 1.5, 2.0, 2.5, 3.0], True, 0),
              Q.WeightsUniformInferableQuantizer(8, [-1.0] * 6, [0.5, 1.0, 1.5, 2.0, 2.5, 3.0], True, 0),
              Q.WeightsPOTInferableQuantizer(4, [2.0], False)):
        codes, s, z = q.quantize_to_codes(x)
        assert codes.dtype in (torch.int8, torch.uint8)
        shape = [-1, 1] if q.per_channel else [1, 1]
        deq = (codes.float() - z.float().reshape(shape)) * s.reshape(shape)
        assert torch.equal(deq, q(x.clone()))
    qa = Q.ActivationUniformInferableQuantizer(8, [-2.5], [3.1])
    codes, s, z = qa.quantize_to_codes(x)
    assert codes.dtype == torch.uint8
    assert torch.equal((codes.float() - z) * torch.tensor(s, dtype=torch.float32), qa(x))


def test_require_hip_switch_refuses_cpu_tensors(monkeypatch):
    q = Q.ActivationSymmetricInferableQuantizer(8, [4.0], True)
    monkeypatch.setenv("MCTQ_REQUIRE_HIP", "1")
    with pytest.raises(RuntimeError, match="MCTQ_REQUIRE_HIP"):
        q(torch.zeros(4))
    monkeypatch.setenv("MCTQ_REQUIRE_HIP", "0")
    assert q(torch.zeros(4)).abs().sum() == 0


def test_torch_compile_traces_through_the_custom_ops():
    lin = torch.nn.Linear(16, 8)
    m = torch.nn.Sequential(
        PytorchQuantizationWrapper(lin, {"weight": Q.WeightsSymmetricInferableQuantizer(8, [1.0] * 8, True, 0)}),
        PytorchActivationQuantizationHolder(Q.ActivationUniformInferableQuantizer(8, [-2.5], [3.1])))
    x = torch.randn(4, 16)
    want = m(x)                                        # (the first call turns requires_grad off on the weights, as the
    torch._dynamo.reset()                              # reference does -- an attribute write dynamo would break on)
    assert torch.equal(torch.compile(m, backend="aot_eager", fullgraph=True)(x), want)
    ex = torch._dynamo.explain(m)(x)
    targets = [str(n.target) for g in ex.graphs for n in g.graph.nodes if n.op == "call_function"]
    assert ex.graph_count == 1 and ex.graph_break_count == 0
    assert "mctq_amd.fq_per_channel" in targets and "mctq_amd.fq_per_tensor" in targets


def test_versioned_reuse_requantizes_only_when_the_weight_changes():
    for q in (Q.WeightsSymmetricInferableQuantizer(8, [1.0, 2.0], True, 0),
              Q.WeightsUniformInferableQuantizer(8, [-1.0], [1.0], False),
              Q.WeightsLUTSymmetricInferableQuantizer(2, [-25.0, 25.0], [1.0], False)):
        w = torch.nn.Parameter(torch.randn(2, 16))
        q.enable_versioned_reuse()
        a = q(w)
        assert q(w) is a                                   # unchanged: the very same tensor
        with torch.no_grad():
            w.neg_()                                       # in-place write bumps the version
        b = q(w)
        assert b is not a and not torch.equal(a, b)
        assert q(w) is b
        other = torch.nn.Parameter(w.detach().clone())
        assert q(other) is not b                           # a different tensor is never served from the cache
        q.disable_versioned_reuse()
        assert q(other) is not q(other)


def test_reference_import_paths_exist():
    """s/mct_quantizers/mct_quantizers_amd/ in an import statement keeps working for the hot-path modules."""
    import importlib
    paths = {
        "common.base_inferable_quantizer": ["BaseInferableQuantizer", "QuantizationTarget", "mark_quantizer", "QuantizerID"],
        "common.quant_info": ["QuantizationMethod"],
        "common.get_quantizers": ["get_inferable_quantizer_class"],
        "common.get_all_subclasses": ["get_all_subclasses"],
        "common.constants": ["EPS", "LUT_VALUES_BITWIDTH", "TRAINING", "LAYER"],
        "logger": ["Logger"],
        "pytorch.quantizer_utils": ["get_working_device", "to_torch_tensor", "fix_range_to_include_zero", "lut_quantizer"],
        "pytorch.quantize_wrapper": ["PytorchQuantizationWrapper"],
        "pytorch.activation_quantization_holder": ["PytorchActivationQuantizationHolder"],
        "pytorch.fln_activation_quantization_holder": ["PytorchFLNActivationQuantizationHolder"],
        "pytorch.preserving_activation_quantization_holder": ["PytorchPreservingActivationQuantizationHolder"],
        "pytorch.load_model": ["pytorch_load_quantized_model"],
        "pytorch.quantizers.base_pytorch_inferable_quantizer": ["BasePyTorchInferableQuantizer"],
        "pytorch.quantizers.weights_inferable_quantizers.weights_lut_pot_inferable_quantizer": ["WeightsLUTPOTInferableQuantizer"],
        "pytorch.quantizers.activation_inferable_quantizers.activation_uniform_inferable_quantizer": ["ActivationUniformInferableQuantizer"],
    }
    for rel, names in paths.items():
        mod = importlib.import_module("mct_quantizers_amd." + rel)
        for n in names:
            assert hasattr(mod, n), (rel, n)


def test_packed_4bit_codes_on_cpu_tensors_match_the_oracle_packing():
    import warnings
    import numpy as np
    import torch
    import mct_quantizers_amd as mq
    from mct_quantizers_amd.hip import ops
    from oracle import oracle_call, mctq_oracle as O
    Q = mq.pytorch_quantizers
    rng = np.random.default_rng(4)
    x = torch.from_numpy((rng.standard_normal((6, 16)) * 1.5).astype(np.float32))
    kw = dict(num_bits=4, threshold=[float(v) for v in rng.uniform(0.5, 3, 6)], per_channel=True, channel_axis=0)
    q = Q.WeightsSymmetricInferableQuantizer(**kw)
    packed, scales, zps = q.quantize_to_codes(x, packed4=True)
    _, idx = oracle_call("WeightsSymmetricInferableQuantizer", kw, x.numpy(), return_index=True)
    assert packed.shape == (6, 8) and np.array_equal(packed.numpy().reshape(-1), O.pack4(idx))
    codes = ops.unpack4(packed, True, (6, 16))
    assert torch.equal(codes.float() * scales.cpu().reshape(-1, 1), q(x))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        a = Q.ActivationUniformInferableQuantizer(num_bits=3, min_range=[-1.0], max_range=[2.0])
    packed, scale, zp = a.quantize_to_codes(x, packed4=True)
    _, idx = oracle_call("ActivationUniformInferableQuantizer", dict(num_bits=3, min_range=[-1.0], max_range=[2.0]),
                         x.numpy(), return_index=True)
    assert np.array_equal(packed.numpy().reshape(-1), O.pack4(idx))
    import pytest
    with pytest.raises(ValueError):
        Q.ActivationSymmetricInferableQuantizer(num_bits=8, threshold=[1.0], signed=True).quantize_to_codes(x, packed4=True)


def test_batched_weight_quantization_on_cpu_tensors_matches_per_layer_calls():
    """pytorch/batching.py on CPU tensors (no batched kernel there: the general route quantizes one by one) --
    same results, same re-quantize-every-forward semantics, clean removal."""
    from mct_quantizers_amd.pytorch.batching import batch_weight_quantization
    torch.manual_seed(0)
    mods = []
    for fin, fout in ((8, 16), (16, 4)):
        lin = torch.nn.Linear(fin, fout)
        mods.append(PytorchQuantizationWrapper(lin, {
            "weight": Q.WeightsSymmetricInferableQuantizer(8, [0.5] * fout, True, 0),
            "bias": Q.WeightsUniformInferableQuantizer(8, [-0.4], [0.6], False)}))
        mods.append(PytorchActivationQuantizationHolder(Q.ActivationUniformInferableQuantizer(8, [-2.5], [3.1])))
    model = torch.nn.Sequential(*mods)
    x = torch.randn(3, 8)
    want = model(x)
    handle = batch_weight_quantization(model)
    assert handle.quantize_now() == 4
    assert torch.equal(model(x), want)
    with torch.no_grad():
        model[0].weight.mul_(0.25)
    changed = model(x)
    handle.remove()
    assert torch.equal(model(x), changed) and not torch.equal(changed, want)
    assert all("_prequantized" not in m.__dict__ for m in model)


def test_tensor_qparams_op_and_fx_routing_on_cpu():
    from mct_quantizers_amd import compat
    from mct_quantizers_amd.hip import ops
    x = torch.randn(5, 9)
    s, z = torch.tensor([0.05]), torch.tensor([3], dtype=torch.int32)
    want = torch.fake_quantize_per_tensor_affine(x, s, z, 0, 255)
    assert torch.equal(ops.fq_per_tensor_tqp(x, s, z, 0, 255), want)
    assert torch.equal(torch.ops.mctq_amd.fq_per_tensor_tqp(x, s, z, 0, 255), want)

    class M(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.register_buffer("s", s.clone())
            self.register_buffer("z", z.clone())

        def forward(self, t):
            return torch.fake_quantize_per_tensor_affine(t, self.s, self.z, 0, 255)

    gm = torch.fx.symbolic_trace(M())
    assert compat.route_fx_graph(gm) == 1
    assert "fq_per_tensor_tqp" in gm.code and torch.equal(gm(x), want)


def test_launch_attributes_invalidate_the_plan_and_survive_pickling():
    import copy
    import pickle
    q = Q.ActivationUniformInferableQuantizer(8, [-2.5], [3.1])
    x = torch.randn(4, 7)
    q.scale = 0.125
    q.zero_point = 9
    assert torch.equal(q(x), torch.fake_quantize_per_tensor_affine(x, 0.125, 9, 0, 255))
    assert q.__dict__["scale"] == 0.125                     # plain instance attributes, as in the reference's pickles
    for clone in (pickle.loads(pickle.dumps(q)), copy.deepcopy(q)):
        assert clone.scale == 0.125 and torch.equal(clone(x), q(x))
    w = Q.WeightsSymmetricInferableQuantizer(8, [1.0, 2.0], True, 0)
    t = torch.randn(2, 5)
    w.zero_points[1] = 4
    assert torch.equal(w(t.clone()), torch.fake_quantize_per_channel_affine(t, w.scales, w.zero_points, 0, -128, 127))
    assert "_plan" not in pickle.loads(pickle.dumps(w)).__getstate__()
    # clamp domains beyond the kernels' float32 bounds (num_bits > 24) are accepted, as in the reference
    # (base_symmetric_inferable_quantizer.py:53-60 hands int64 bounds to ATen), and run ATen's own operator
    for nb in (25, 31):
        wide = Q.WeightsSymmetricInferableQuantizer(nb, [1.0, 3.0], True, 0)
        big = torch.randn(2, 5) * 1e6
        assert torch.equal(wide(big.clone()), torch.fake_quantize_per_channel_affine(
            big, wide.scales, wide.zero_points, 0, -2 ** (nb - 1), 2 ** (nb - 1) - 1))
        au = Q.ActivationUniformInferableQuantizer(nb, [-1.0], [3.0])
        assert torch.equal(au(big), torch.fake_quantize_per_tensor_affine(big, au.scale, au.zero_point, 0, 2 ** nb - 1))
        assert au.__dict__["_plan"] is False and wide.__dict__["_plan"] is None


def test_jit_trace_on_cpu_records_the_reference_nodes():
    h = PytorchActivationQuantizationHolder(Q.ActivationUniformInferableQuantizer(8, [-2.5], [3.1]))
    x = torch.randn(3, 5)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        tr = torch.jit.trace(h, x, check_trace=False)
    assert any("fake_quantize_per_tensor_affine" in n.kind() for n in tr.graph.nodes())
    assert torch.equal(tr(x * 2), h(x * 2))


def test_lut_quantizer_accepts_a_negative_channel_axis():
    from mct_quantizers_amd.pytorch.quantizer_utils import lut_quantizer
    x = torch.randn(3, 4, 5)
    lut = torch.tensor([-100.0, 0.0, 60.0, 127.0])
    thr = torch.tensor([1.0, 2.0, 0.5, 4.0, 1.5])
    a = lut_quantizer(x, lut, True, thr, 8, 1e-8, per_channel=True, channel_axis=-1, input_rank=3)
    b = lut_quantizer(x, lut, True, thr, 8, 1e-8, per_channel=True, channel_axis=2, input_rank=3)
    assert torch.equal(a, b)


def test_half_activation_lut_on_cpu_tensors_matches_the_reference_fixtures():
    """CPU tensors take the torch op chain of the reference; float16 / bfloat16 activations with wide codebooks incl. the
    configuration torch refuses (tests/golden/cases_half_bounds.*, generated from the reference)."""
    import json
    import os
    import warnings
    from conftest import GOLDEN, bits_equal
    with open(os.path.join(GOLDEN, "cases_half_bounds.json")) as f:
        cases = json.load(f)["cases"]
    arrays = np.load(os.path.join(GOLDEN, "cases_half_bounds.npz"))
    for c in cases:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            q = Q.ActivationLutPOTInferableQuantizer(**c["kwargs"])
        x = torch.from_numpy(arrays[c["id"] + "_x"]).to(getattr(torch, c["in_dtype"]))
        if "error" in c:
            with pytest.raises(RuntimeError) as e:
                q(x)
            assert str(e.value) == c["error"]
            continue
        y = q(x)
        assert str(y.dtype) == "torch." + c["out_dtype"] and bits_equal(y.float().numpy(), arrays[c["id"] + "_y"]), c["id"]


def test_lut_quantizers_follow_attribute_assignment_on_cpu():
    lut = [-128.0, -64.0, -20.0, -5.0, 0.0, 5.0, 20.0, 64.0]
    x = torch.randn(3, 50) * 3
    q = Q.ActivationLutPOTInferableQuantizer(3, lut, [4.0], True)
    a = q(x)
    q.threshold = 8.0
    assert torch.equal(q(x), Q.ActivationLutPOTInferableQuantizer(3, lut, [8.0], True)(x)) and not torch.equal(q(x), a)
    q.eps = 0.5
    assert torch.equal(q(x), Q.ActivationLutPOTInferableQuantizer(3, lut, [8.0], True, eps=0.5)(x))
    w = Q.WeightsLUTSymmetricInferableQuantizer(3, lut, [1.0, 2.0, 0.5], True, 0, 2)
    w(x.clone())
    w._threshold_torch = torch.tensor([3.0, 0.25, 1.5])
    assert torch.equal(w(x.clone()), Q.WeightsLUTSymmetricInferableQuantizer(3, lut, [3.0, 0.25, 1.5], True, 0, 2)(x.clone()))
    import pickle
    w2 = pickle.loads(pickle.dumps(w))
    assert torch.equal(w2(x.clone()), w(x.clone())) and not w2.__dict__.get("_stale")


def test_dotted_module_paths_of_the_reference_resolve():
    """Code written against the reference reaches classes through its package attributes without importing the
    sub-modules (its top level imports them all): the same dotted paths resolve here, to the same objects."""
    import mct_quantizers_amd as m
    pq = m.pytorch_quantizers
    assert pq.weights_inferable_quantizers.weights_symmetric_inferable_quantizer.WeightsSymmetricInferableQuantizer \
        is pq.WeightsSymmetricInferableQuantizer
    assert pq.activation_inferable_quantizers.activation_lut_pot_inferable_quantizer.ActivationLutPOTInferableQuantizer \
        is pq.ActivationLutPOTInferableQuantizer
    assert pq.base_pytorch_inferable_quantizer.BasePyTorchInferableQuantizer is pq.BasePyTorchInferableQuantizer
    assert m.pytorch.quantize_wrapper.PytorchQuantizationWrapper is m.PytorchQuantizationWrapper
    assert m.pytorch.activation_quantization_holder.PytorchActivationQuantizationHolder is m.PytorchActivationQuantizationHolder
    assert m.common.base_inferable_quantizer.BaseInferableQuantizer is m.BaseInferableQuantizer
    assert m.common.get_quantizers.get_inferable_quantizer_class is m.get_inferable_quantizer_class
    with pytest.raises(AttributeError):
        m.pytorch.no_such_module


def test_bench_emits_measured_traffic_only_for_the_build_the_counters_were_taken_on(tmp_path):
    """roofline.traffic is tied to the library's build id, not to the kernel variant's name alone."""
    import json
    import bench_dist
    rec = {"cfg2": {"variant": "rows_kernel<AffineOp,in4B,out4B,U=4,NT=1>", "build_id": "0123456789abcdef", "git_head": "abc",
                    "hbm_bytes_per_launch": 134539156.0}}
    path = tmp_path / "pmc_traffic.json"
    path.write_text(json.dumps(rec))
    v = rec["cfg2"]["variant"]
    ok = bench_dist.traffic_fields(str(path), "cfg2", v, "0123456789abcdef")
    assert ok["traffic"] == 134539156.0 and ok["traffic_build_id_mismatch"] is False
    other = bench_dist.traffic_fields(str(path), "cfg2", v, "fedcba9876543210")            # same variant, another build
    assert other["traffic"] is None and other["traffic_build_id_mismatch"] is True
    assert other["traffic_recorded_for_another_build"] == 134539156.0 and "0123456789abcdef" in other["traffic_source"]
    rec["cfg2"].pop("build_id")                                                             # a record from before build ids
    path.write_text(json.dumps(rec))
    old = bench_dist.traffic_fields(str(path), "cfg2", v, "0123456789abcdef")
    assert old["traffic"] is None and old["traffic_build_id_mismatch"] is True
    stale = bench_dist.traffic_fields(str(path), "cfg2", "rows_kernel<AffineOp,in4B,out4B,U=8,NT=1>", "0123456789abcdef")
    assert stale["traffic"] is None and "stale" in stale["traffic_source"] and "traffic_build_id_mismatch" not in stale
    assert bench_dist.traffic_fields(str(path), "cfg9", v, "x")["traffic"] is None
    assert bench_dist.traffic_fields(str(tmp_path / "none.json"), "cfg2", v, "x")["traffic_source"].endswith("missing")


def test_split_rows_cuts_a_tensor_into_row_blocks_below_the_launch_limit():
    """hip/ops.py::_split_rows (include/mctq_hip.h "Size limit"): whole outer slices when a slice fits the limit,
    runs of channel rows of one slice otherwise; per-channel parameters sliced alike; the result keeps x's strides."""
    from mct_quantizers_amd.hip import ops
    torch.manual_seed(0)
    calls = []

    def call(xp, ps, ax):
        assert xp.dim() == 3 and ax == 1 and xp.is_contiguous() and ps[0].numel() == xp.shape[1]
        calls.append(tuple(xp.shape))
        return xp * ps[0].view(1, -1, 1) + ps[1].view(1, -1, 1)
    for shape, axis, limit in (((5, 7, 3), 1, 50), ((5, 7, 3), 1, 8), ((5, 7, 3), 0, 40), ((6, 4), 1, 9), ((2, 3, 4, 5), 1, 31),
                               ((2, 3, 4, 5), 3, 17), ((5, 7, 3), 1, 10 ** 6)):
        for permuted in (False, True):
            x = torch.randn(*shape)
            ax = axis
            if permuted and x.dim() == 4:
                x = x.to(memory_format=torch.channels_last)
            c = x.shape[ax]
            a, b = torch.randn(c), torch.randn(c)
            view = [1] * x.dim()
            view[ax] = -1
            calls.clear()
            y = ops._split_rows(x, ax, (a, b), torch.float32, call, limit=limit)
            assert y.shape == x.shape and y.stride() == x.stride()
            assert torch.equal(y, x * a.view(view) + b.view(view)), (shape, axis, limit, permuted)
            assert all(k * ch * inner <= max(limit, inner) for k, ch, inner in calls) and len(calls) >= 1
    with pytest.raises(NotImplementedError):
        ops._split_rows(torch.randn(2, 3, 40), 1, (torch.randn(3),), torch.float32, lambda xp, ps, ax: xp, limit=16)
