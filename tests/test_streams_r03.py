"""CapturedStream (pytorch/graphs.py): a fixed-shape stream of activation batches through a holder, `depth` batches per
replay -- one fused batched launch for the affine activation quantizers, one hipGraph otherwise.  Results are compared
with the oracle; everything that does not fit the captured shape must fall back to the eager calls
(reference call site: pytorch/activation_quantization_holder.py:43-53)."""
import numpy as np
import pytest
import torch

from conftest import bits_equal

pytestmark = pytest.mark.gpu


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
