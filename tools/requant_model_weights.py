"""Re-quantizing ALL weights of a model, as every forward of an MCT-exported model does
(reference pytorch/quantize_wrapper.py:228-240): one quantizer call per layer vs ONE batched launch
(ops.fq_batched -> mctq_fq_batched).  Weight shapes of ResNet-50 (54 tensors, 25.5 M parameters) and of a
16 x Linear(4096) stack; per-channel symmetric 8 bit along axis 0.  Outputs are compared bit for bit."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mct_quantizers_amd as mq
from mct_quantizers_amd.hip import native, ops
Q = mq.pytorch_quantizers

def resnet50_shapes():
    from mct_quantizers_amd import workloads
    return workloads.model_weight_shapes("resnet50")

LUT16 = [-128.0, -96.0, -64.0, -40.0, -24.0, -12.0, -5.0, 0.0, 5.0, 12.0, 24.0, 40.0, 64.0, 96.0, 120.0, 127.0]


def run_lut(name, shapes, reps=200):
    """The same lists with LUT weights quantizers (16-entry codebook, per channel): per-layer calls vs the pre-packed plan
    (one table launch, mctq_lutt_batch_run) vs what the reference runs (its op chain, ~10 kernels per layer; timed on the
    four smallest layers only -- it materialises N x 16 temporaries)."""
    torch.manual_seed(0)
    ws = [torch.randn(s, device="cuda") * 0.05 for s in shapes]
    qs = [Q.WeightsLUTSymmetricInferableQuantizer(4, list(LUT16), [float(v) + 1e-6 for v in w.reshape(w.shape[0], -1).abs().amax(dim=1)],
                                                  True, 0, w.dim()) for w in ws]
    per_layer = lambda: [q(w) for w, q in zip(ws, qs)]
    outs = [torch.empty(w.shape, dtype=torch.float32, device="cuda") for w in ws]
    items = []
    for w, q, o in zip(ws, qs, outs):
        it = q.batch_item_lut(w)
        items.append(it[:2] + (o,) + it[3:])
    plan = native.fast().BatchPlan(items)
    a = per_layer(); plan(); torch.cuda.synchronize()
    same = all(torch.equal(x, o) for x, o in zip(a, outs))
    nbytes = sum(w.numel() for w in ws) * 8
    print(f"{name} (LUT): {len(ws)} tensors, {nbytes / 8e6:.1f} M elements, bit-equal={same}")
    for label, f in (("this package, one call per layer", per_layer), ("  pre-packed plan + persistent outputs, ONE launch", plan)):
        for _ in range(10): f()
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(reps): f()
        th = time.perf_counter()
        torch.cuda.synchronize()
        v, h = (time.perf_counter() - t) / reps * 1e6, (th - t) / reps * 1e6
        print(f"    {label:48s} {v:9.1f} us per model (host {h:7.1f})   {nbytes / v / 1e3:7.0f} GB/s algorithmic")


def run(name, shapes, reps=200):
    torch.manual_seed(0)
    ws = [torch.randn(s, device="cuda") * 0.05 for s in shapes]
    qs = [Q.WeightsSymmetricInferableQuantizer(8, [float(v) + 1e-6 for v in w.reshape(w.shape[0], -1).abs().amax(dim=1)], True, 0) for w in ws]
    aten = lambda: [torch.fake_quantize_per_channel_affine(w, q.scales, q.zero_points, 0, -128, 127) for w, q in zip(ws, qs)]
    per_layer = lambda: [q(w) for w, q in zip(ws, qs)]
    batched = lambda: ops.fq_batched([q.batch_item(w) for w, q in zip(ws, qs)])
    outs = [torch.empty_like(w) for w in ws]
    plan = native.fast().BatchPlan([q.batch_item(w)[:1] + (o,) + q.batch_item(w)[1:] for w, q, o in zip(ws, qs, outs)])
    planned = lambda: plan()
    a, b, c = aten(), per_layer(), batched()
    planned(); torch.cuda.synchronize()
    same = all(torch.equal(x, y) and torch.equal(x, z) and torch.equal(x, o) for x, y, z, o in zip(a, b, c, outs))
    nbytes = sum(w.numel() for w in ws) * 8
    res = {}
    for label, f in (("ATen ops (what the reference runs)", aten), ("this package, one call per layer", per_layer), ("this package, ONE batched launch", batched),
                     ("  same, pre-packed plan + persistent outputs", planned)):
        for _ in range(10): f()
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(reps): f()
        th = time.perf_counter()
        torch.cuda.synchronize()
        res[label] = ((time.perf_counter() - t) / reps * 1e6, (th - t) / reps * 1e6)
    print(f"{name}: {len(ws)} tensors, {nbytes / 8e6:.1f} M elements, bit-equal={same}")
    for k, (v, h) in res.items():
        print(f"    {k:48s} {v:9.1f} us per model (host {h:7.1f})   {nbytes / v / 1e3:7.0f} GB/s algorithmic")

if __name__ == "__main__":
    run("ResNet-50 weights", resnet50_shapes())
    run("MobileNet-like pointwise/depthwise stack", [(c, 1, 3, 3) for c in (32, 64, 128, 128, 256, 256, 512, 512, 512, 512, 512, 512, 1024)] +
        [(co, ci, 1, 1) for ci, co in ((32, 64), (64, 128), (128, 128), (128, 256), (256, 256), (256, 512), (512, 512), (512, 512), (512, 512), (512, 512), (512, 512), (512, 1024), (1024, 1024))])
    run("16 x Linear(4096,4096)", [(4096, 4096)] * 16, reps=50)
    run_lut("ResNet-50 weights", resnet50_shapes())
    run_lut("16 x Linear(4096,4096)", [(4096, 4096)] * 16, reps=50)
