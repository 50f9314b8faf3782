"""GPU: the batched multi-tensor launch through the public paths -- mixed shapes / axes / dtypes against the oracle, a
wrapped model's weights per forward (reference pytorch/quantize_wrapper.py:228-240) with and without persistent buffers.
(Raw ABI, table packer, plans, LUT batches: test_batched.py.)"""
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


@torch.no_grad()      # inference forwards: a forward autograd may record gets fresh tensors, not the plan (pytorch/batching.py)
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
