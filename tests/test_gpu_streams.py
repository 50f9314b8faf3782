"""GPU: graphs and streams.  CapturedStream (pytorch/graphs.py): a fixed-shape stream of activation batches through a holder, `depth` batches per
replay -- one fused batched launch for the affine activation quantizers, one hipGraph otherwise.  Results are compared
with the oracle; everything that does not fit the captured shape must fall back to the eager calls
(reference call site: pytorch/activation_quantization_holder.py:43-53)."""
import hashlib
import json
import os
import warnings

import numpy as np
import pytest
import torch

from conftest import GOLDEN, bits_equal, finite_equal, first_mismatch, load_json  # noqa: F401

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


SPECS = {
    "uniform": ("ActivationUniformInferableQuantizer", dict(num_bits=8, min_range=[-2.5], max_range=[3.1])),
    "symmetric": ("ActivationSymmetricInferableQuantizer", dict(num_bits=4, threshold=[2.0], signed=True)),
    "lut": ("ActivationLutPOTInferableQuantizer", dict(num_bits=3, lut_values=[-128.0, -64.0, -20.0, -5.0, 0.0, 5.0, 20.0, 64.0],
                                                       threshold=[4.0], signed=True)),
}


def _holder(kind):
    import warnings
    import mct_quantizers_amd as mq
    cls, kw = SPECS[kind]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        q = getattr(mq.pytorch_quantizers, cls)(**kw)
    return mq.PytorchActivationQuantizationHolder(q).cuda(), q


def _oracle(kind, x):
    from oracle import oracle_call
    cls, kw = SPECS[kind]
    return oracle_call(cls, kw, x)


@pytest.mark.parametrize("kind,mode", [("uniform", "fused"), ("symmetric", "fused"), ("lut", "fused"), ("lut", "graph"),
                                       ("uniform", "graph")])
def test_stream_results_are_the_oracles_and_misfits_fall_back_to_eager(kind, mode):
    from mct_quantizers_amd.hip import native
    if native.fast() is None and mode == "fused":
        pytest.skip("fused streams need the compiled binding")
    torch.manual_seed(2)
    holder, q = _holder(kind)
    depth = 6
    example = torch.randn(2, 3, 20, 33, device="cuda") * 2
    st = holder.capture_stream(example, depth=depth, mode="auto" if mode == "fused" else "graph")
    assert st.mode == mode and (kind != "lut" or st.outputs[0].dtype == torch.float32)
    batches = [torch.randn_like(example) * (1 + i) for i in range(depth)]
    outs = st(batches)
    torch.cuda.synchronize()
    for b, y in zip(batches, outs):
        assert bits_equal(y.cpu().numpy(), _oracle(kind, b.cpu().numpy()))
    # zero-copy use: the producer writes into the static inputs, run() replays
    for i, x in enumerate(st.inputs):
        x.copy_(batches[(i + 1) % depth])
    outs = st.run()
    for i, y in enumerate(outs):
        assert bits_equal(y.cpu().numpy(), _oracle(kind, batches[(i + 1) % depth].cpu().numpy()))
    kernel = native.last_launch()
    assert (("batched_kernel<table>" in kernel) or ("batched_lut_kernel<table>" in kernel)) == (mode == "fused"), kernel
    # another shape, another count, another dtype: eager calls, same bits
    for odd in ([torch.randn(5, 7, device="cuda") for _ in range(depth)], batches[:3],
                [b.half() for b in batches] if kind != "lut" else batches[:1]):
        got = st(odd)
        assert len(got) == len(odd)
        for b, y in zip(odd, got):
            assert torch.equal(y, holder(b))
    st.release()
    assert torch.equal(st(batches)[0], holder(batches[0]))          # released: eager


def test_fused_stream_notices_changed_quantizer_parameters():
    from mct_quantizers_amd.hip import native
    if native.fast() is None:
        pytest.skip("fused streams need the compiled binding")
    holder, q = _holder("uniform")
    example = torch.randn(4, 50, device="cuda")
    st = holder.capture_stream(example, depth=3)
    assert st.mode == "fused"
    st.run()
    q.scale = q.scale * 2.0                                           # the reference reads its attributes on every call
    outs = st.run()
    assert st.mode == "eager"
    for x, y in zip(st.inputs, outs):
        assert torch.equal(y, torch.fake_quantize_per_tensor_affine(x, q.scale, q.zero_point, 0, 255))


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


def test_captures_survive_an_aggressive_garbage_collector(lib):
    """Captured objects form reference cycles with their models; if the cyclic collector frees an older hipGraph WHILE a
    newer capture is recording, the runtime call in its destructor aborts the process (seen in the accelerate fuzz,
    profiles/r04/fuzz_soak.log: the accelerate fuzz at seed 21 reproduced it before pytorch/graphs.py:
    no_gc_while_capturing).  A stress of the same pattern, not a deterministic reproduction (when the collector strikes
    depends on allocation counts): captured forwards, auto-captured models and graph-mode streams are built, used and
    dropped WITHOUT release() in a loop, half of the rounds with the collector at its most eager, half with it lazy."""
    import gc
    import mct_quantizers_amd as mq
    from mct_quantizers_amd.hip import native
    Q = mq.pytorch_quantizers

    def model():
        lin = torch.nn.Linear(32, 16).cuda()
        thr = [float(v) for v in lin.weight.detach().abs().amax(dim=1)]
        return torch.nn.Sequential(
            mq.PytorchQuantizationWrapper(lin, {"weight": Q.WeightsSymmetricInferableQuantizer(8, thr, True, 0)}),
            mq.PytorchActivationQuantizationHolder(Q.ActivationUniformInferableQuantizer(8, [-2.5], [3.1]))).eval()

    x = torch.randn(4, 32, device="cuda")
    old = gc.get_threshold()
    try:
        for i in range(16):
            gc.set_threshold(*((1, 1, 1) if i % 2 else (700, 10, 10)))
            m = mq.accelerate(model(), capture=True)                         # model <-> AutoCapture cycle, graphs inside
            with torch.no_grad():
                want = m(x)
                for _ in range(3):
                    assert torch.equal(m(x), want)
            if native.fast() is not None:                                      # (the replay reads the pre-packed plan's buffers: compiled binding)
                assert m.__dict__["_mctq_auto_capture"]._graphs
            cf = mq.capture_forward(model(), x)                              # CapturedForward
            assert cf(x).shape == (4, 16)
            lutq = Q.ActivationLutPOTInferableQuantizer(3, [-128.0, -64.0, -20.0, -5.0, 0.0, 5.0, 20.0, 64.0], [4.0], True)
            st = mq.capture_stream(lutq, x, depth=3, mode="graph")           # CapturedStream, hipGraph mode
            st.run()
            del m, cf, st                                                    # dropped WITHOUT release(): garbage with graphs
    finally:
        gc.set_threshold(*old)
        gc.collect()
