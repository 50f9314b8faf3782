"""The mctq_amd::* library ops carry ATen's straight-through backward.

torch.fake_quantize_per_{tensor,channel}_affine -- the operators at the reference's call sites
(weights_symmetric_inferable_quantizer.py:139-151, activation_uniform_inferable_quantizer.py:124) -- pass the gradient where
the clamp index lies inside [quant_min, quant_max]; the ops that replace them in fx / torch.compile graphs must give the
same x.grad, and the LUT ops' results are not differentiable (argmin + gather, quantizer_utils.py:131-137).
"""
import warnings

import pytest
import torch


def _grads(device, dtype=torch.float32):
    import mct_quantizers_amd  # noqa: F401  (registers the ops)
    torch.manual_seed(7)
    out = []
    x0 = (torch.randn(6, 8, 5, device=device) * 3).to(dtype)
    g = torch.randn(6, 8, 5, device=device).to(dtype)
    s = (torch.rand(8, device=device) * 0.1 + 0.01)
    z = torch.randint(-3, 4, (8,), dtype=torch.int32, device=device)
    st, zt = torch.tensor([0.07], device=device), torch.tensor([2], dtype=torch.int32, device=device)
    cases = [
        ("per_tensor", lambda x: torch.ops.mctq_amd.fq_per_tensor(x, 0.05, 3, -20, 20),
         lambda x: torch.fake_quantize_per_tensor_affine(x, 0.05, 3, -20, 20)),
        ("per_tensor_tqp", lambda x: torch.ops.mctq_amd.fq_per_tensor_tqp(x, st, zt, 0, 15),
         lambda x: torch.fake_quantize_per_tensor_affine(x, st, zt, 0, 15)),
        ("per_channel", lambda x: torch.ops.mctq_amd.fq_per_channel(x, s, z, 1, -8, 7),
         lambda x: torch.fake_quantize_per_channel_affine(x, s, z, 1, -8, 7)),
    ]
    for name, ours, aten in cases:
        xa, xb = x0.clone().requires_grad_(), x0.clone().requires_grad_()
        ya, yb = ours(xa), aten(xb)
        (ya * g).sum().backward()
        (yb * g).sum().backward()
        out.append((name, ya.detach(), yb.detach(), xa.grad, xb.grad))
    return out


def test_affine_ops_have_atens_straight_through_backward_on_cpu():
    with warnings.catch_warnings():
        warnings.simplefilter("error")                       # no "autograd kernel was not registered" fallback warning
        for name, ya, yb, ga, gb in _grads("cpu"):
            assert torch.equal(ya, yb), name
            assert torch.equal(ga, gb), name
            assert 0 < int((ga != 0).sum()) < ga.numel(), name   # the mask is neither empty nor full: clipped elements exist


def test_lut_ops_are_not_differentiable():
    import mct_quantizers_amd  # noqa: F401
    x = torch.randn(4, 6, requires_grad=True)
    lut = torch.tensor([-100.0, -20.0, 0.0, 30.0, 127.0])
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        y = torch.ops.mctq_amd.lut_per_tensor(x, lut, 4.0, 4.0, 128.0, -128.0, 127.0)
        yc = torch.ops.mctq_amd.lut_per_channel(x, lut, torch.tensor([1.0, 2.0, 3.0, 4.0]), 1e-8, 0, 128.0, -128.0, 127.0)
    assert not y.requires_grad and not yc.requires_grad      # as the reference's chain: argmin + gather cut the graph


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_affine_ops_backward_equals_aten_on_the_gpu(dtype):
    from mct_quantizers_amd.hip import native
    native.load()
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        for name, ya, yb, ga, gb in _grads("cuda", dtype):
            assert torch.equal(ya, yb), name
            assert torch.equal(ga, gb), name
    assert "kernel" in native.last_launch()


@pytest.mark.gpu
def test_compiled_model_with_grad_enabled_inputs_raises_no_autograd_warning():
    """The torch.compile GPU test of round 2 carried PyTorch's 'autograd kernel was not registered' warning: with the
    registration the same graph compiles and runs with warnings as errors, forward bits unchanged."""
    import mct_quantizers_amd as mq
    Q = mq.pytorch_quantizers
    torch.manual_seed(3)
    lin = torch.nn.Linear(64, 24).cuda()
    thr = [0.5 + 0.05 * i for i in range(24)]
    m = torch.nn.Sequential(
        mq.PytorchActivationQuantizationHolder(Q.ActivationUniformInferableQuantizer(8, [-2.5], [3.1])),
        mq.PytorchQuantizationWrapper(lin, {"weight": Q.WeightsSymmetricInferableQuantizer(8, thr, True, 0)}),
        mq.PytorchActivationQuantizationHolder(Q.ActivationLutPOTInferableQuantizer(3, [-128.0, -64.0, -20.0, -5.0, 0.0, 5.0, 20.0, 64.0], [4.0], True)),
    ).cuda()
    x = torch.randn(32, 64, device="cuda", requires_grad=True)
    want = m(x.detach())
    torch._dynamo.reset()
    with warnings.catch_warnings():
        warnings.filterwarnings("error", message=".*autograd kernel was not registered.*")
        got = torch.compile(m, backend="aot_eager", fullgraph=True)(x)
        torch.cuda.synchronize()
    assert torch.equal(got.detach(), want)
