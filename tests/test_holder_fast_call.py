"""The eager call of an activation holder (reference: pytorch/activation_quantization_holder.py:43-53, forward =
``self.activation_holder_quantizer(inputs)``) as ONE C call (compiled binding: HolderCall) -- and every situation in
which it must stand aside: hooks, a swapped quantizer, an assigned parameter, the bypass switch, traces, pickles."""
import copy
import pickle
import warnings

import numpy as np
import pytest
import torch

import mct_quantizers_amd as mq
from conftest import bits_equal

Q = mq.pytorch_quantizers
gpu = pytest.mark.gpu


def _oracle(kw, x_np, cls="ActivationUniformInferableQuantizer", in_dtype="float32"):
    from oracle import oracle_call
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return oracle_call(cls, kw, x_np, in_dtype=in_dtype)


@pytest.fixture
def compiled_binding():
    from mct_quantizers_amd.hip import native
    if native.fast() is None:
        pytest.skip("compiled binding not in use (MCTQ_BINDING=ctypes / MCTQ_ROCTX=1): holders take the Python path")


def test_cpu_tensors_never_build_the_fast_call():
    h = mq.PytorchActivationQuantizationHolder(Q.ActivationUniformInferableQuantizer(8, [-2.5], [3.1]))
    x = torch.randn(2, 3, 8, 8)
    y = h(x)
    assert h.__dict__.get("_fast_call") is None
    assert torch.equal(y, h.activation_holder_quantizer(x))
    clone = pickle.loads(pickle.dumps(h))
    assert torch.equal(clone(x), y)


@gpu
def test_fast_call_is_built_on_first_use_and_matches_the_oracle(compiled_binding):
    kw = dict(num_bits=8, min_range=[-2.5], max_range=[3.1])
    h = mq.PytorchActivationQuantizationHolder(Q.ActivationUniformInferableQuantizer(**kw))
    x_np = (np.random.default_rng(0).standard_normal((2, 3, 32, 32)) * 2).astype(np.float32)
    x = torch.from_numpy(x_np).cuda()
    want = _oracle(kw, x_np)
    y0 = h(x)                                            # slow path, builds the C call
    fast = h.__dict__.get("_fast_call")
    assert fast is not None and type(fast).__name__ == "HolderCall"
    for _ in range(3):
        y = h(x)
        assert y is not y0 and bits_equal(y.cpu().numpy(), want)
    assert bits_equal(y0.cpu().numpy(), want)
    assert fast(x.cpu()) is NotImplemented and fast("no tensor") is NotImplemented
    xh = x.half()                                        # another storage type through the same C call
    assert bits_equal(h(xh).float().cpu().numpy(), _oracle(kw, xh.float().cpu().numpy(), in_dtype="float16"))


@gpu
def test_fast_call_follows_parameter_assignment_and_quantizer_swap(compiled_binding):
    kw = dict(num_bits=8, min_range=[-2.5], max_range=[3.1])
    q = Q.ActivationUniformInferableQuantizer(**kw)
    h = mq.PytorchActivationQuantizationHolder(q)
    x = torch.randn(4, 16, 16, device="cuda") * 2
    h(x); h(x)
    first = h.__dict__["_fast_call"]
    q.scale = 0.5                                        # the reference reads the attribute on every call
    want = torch.fake_quantize_per_tensor_affine(x, 0.5, q.zero_point, 0, 255)
    assert torch.equal(h(x), want)
    assert torch.equal(h(x), want) and h.__dict__["_fast_call"] is not first      # re-made for the new plan
    q2 = Q.ActivationSymmetricInferableQuantizer(4, [2.0], True)
    h.activation_holder_quantizer = q2
    assert torch.equal(h(x), q2(x)) and torch.equal(h(x), q2(x))
    assert h.__dict__["_fast_key"][0] is q2


@gpu
def test_hooks_bypass_and_traces_take_the_module_path(compiled_binding):
    q = Q.ActivationUniformInferableQuantizer(8, [-2.5], [3.1])
    h = mq.PytorchFLNActivationQuantizationHolder(q, quantization_bypass=False)
    x = torch.randn(2, 3, 16, 16, device="cuda")
    want = q(x)
    h(x); assert h.__dict__.get("_fast_call") is not None
    seen = []
    handle = h.register_forward_hook(lambda m, i, o: seen.append(1))
    assert torch.equal(h(x), want) and seen == [1]
    handle.remove()
    assert torch.equal(h(x), want) and seen == [1]
    g = torch.nn.modules.module.register_module_forward_pre_hook(lambda m, i: seen.append(2))
    h(x)
    g.remove()
    assert seen == [1, 2]
    h.quantization_bypass = True
    assert h(x) is x
    h.quantization_bypass = False
    assert torch.equal(h(x), want)
    gm = torch.fx.symbolic_trace(h)
    assert torch.equal(gm(x), want)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        traced = torch.jit.trace(h, x, check_trace=False)
    assert any("fake_quantize" in n.kind() for n in traced.inlined_graph.nodes())
    assert torch.equal(traced(x), want)
    clone = copy.deepcopy(h)
    assert clone.__dict__.get("_fast_call") is None and torch.equal(clone(x), want)
    clone = pickle.loads(pickle.dumps(h))
    assert torch.equal(clone(x), want) and torch.equal(clone(x), want)
    xg = x.clone().requires_grad_(True)                  # grads: the reference runs under no_grad -> no graph
    assert not h(xg).requires_grad


@gpu
def test_lut_and_foreign_quantizers_keep_the_python_path():
    lut = [-128.0, -64.0, -20.0, -5.0, 0.0, 5.0, 20.0, 64.0]
    h = mq.PytorchActivationQuantizationHolder(Q.ActivationLutPOTInferableQuantizer(3, lut, [4.0], True))
    x = torch.randn(2, 3, 16, 16, device="cuda")
    y = h(x)
    assert h.__dict__.get("_fast_call") is None and torch.equal(h(x), y)

    class Mine(mq.BaseInferableQuantizer):
        def __call__(self, t):
            return t * 2
    h2 = mq.PytorchActivationQuantizationHolder(Mine())
    assert torch.equal(h2(x), x * 2) and h2.__dict__.get("_fast_call") is None
