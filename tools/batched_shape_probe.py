"""Where does the batched weight launch (mctq_fq_batched) lose time?  Per distinct weight shape of ResNet-50 (and a few
Linear shapes): K copies of that shape in ONE batched launch vs the dedicated per-channel kernel on the same bytes
viewed as [outer = K, C, inner] (one call of mctq_fq_per_channel), both through pre-packed plans / the raw binding so
that host time is out of the picture; event-timed, buffers rotated over >= 3 sets (cold).  Then the whole model's list.
Outputs are compared bit for bit."""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mct_quantizers_amd as mq  # noqa: E402
from mct_quantizers_amd.hip import native, ops  # noqa: E402
from requant_model_weights import resnet50_shapes  # noqa: E402

Q = mq.pytorch_quantizers
fast = native.fast()


def timed(fs, reps):
    for f in fs:
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        fs[i % len(fs)]()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def probe_shape(shape, k, sets=3, reps=60):
    c = shape[0]
    inner = 1
    for s in shape[1:]:
        inner *= s
    torch.manual_seed(1)
    scales = (torch.rand(c, device="cuda") * 0.01 + 0.001)
    plans, singles, check = [], [], None
    for s in range(sets):
        big = torch.randn((k,) + tuple(shape), device="cuda") * 0.05
        out_b = torch.empty_like(big)
        out_s = torch.empty_like(big)
        items = [(big[i], out_b[i], scales, None, 0, -128, 127) for i in range(k)]
        plans.append(fast.BatchPlan(items))
        v, o = big.view(k * c, inner), out_s.view(k * c, inner)
        sc = scales.repeat(k)
        singles.append((v, o, sc))
        if s == 0:
            check = (out_b, out_s, v, sc)
    fs_b = [p for p in plans]

    def mk(v, o, sc):
        lib = native.load()
        st = torch.cuda.current_stream().cuda_stream
        return lambda: lib.mctq_fq_per_channel(v.data_ptr(), o.data_ptr(), 1, v.shape[0], v.shape[1], native.DT_F32,
                                               sc.data_ptr(), None, -128, 127, st)
    fs_s = [mk(*t) for t in singles]
    tb = timed(fs_b, reps)
    kern_b = native.last_launch()
    ts = timed(fs_s, reps)
    kern_s = native.last_launch()
    same = bool(torch.equal(check[0], check[1]))
    nbytes = k * c * inner * 8
    print(f"  {str(tuple(shape)):22s} inner {inner:5d} x{k:3d} {nbytes / 2**20:7.1f} MiB  batched {tb:8.1f} us {nbytes / tb / 1e3:6.0f} GB/s"
          f"   dedicated {ts:8.1f} us {nbytes / ts / 1e3:6.0f} GB/s [{kern_s}]  equal={same}", flush=True)
    return kern_b


def whole_model(name, shapes, sets, reps=100):
    torch.manual_seed(0)
    plans, first = [], None
    for s in range(sets):
        ws = [torch.randn(sh, device="cuda") * 0.05 for sh in shapes]
        qs = [Q.WeightsSymmetricInferableQuantizer(8, [float(v) + 1e-6 for v in w.reshape(w.shape[0], -1).abs().amax(dim=1)], True, 0)
              for w in ws]
        outs = [torch.empty_like(w) for w in ws]
        plans.append(fast.BatchPlan([q.batch_item(w)[:1] + (o,) + q.batch_item(w)[1:] for w, q, o in zip(ws, qs, outs)]))
        if s == 0:
            first = (ws, qs, outs)
    t = timed(plans, reps)
    ws, qs, outs = first
    plans[0]()
    torch.cuda.synchronize()
    same = all(torch.equal(o, torch.fake_quantize_per_channel_affine(w, q.scales, q.zero_points, 0, -128, 127)) for w, q, o in zip(ws, qs, outs))
    nbytes = sum(w.numel() for w in ws) * 8
    print(f"{name}: {len(ws)} tensors {nbytes / 2**20:.1f} MiB, {sets} buffer set(s): {t:8.1f} us  {nbytes / t / 1e3:6.0f} GB/s  "
          f"== ATen: {same}  [{native.last_launch()}]", flush=True)


if __name__ == "__main__":
    shapes = resnet50_shapes()
    distinct = {}
    for s in shapes:
        distinct[s] = distinct.get(s, 0) + 1
    print("per shape (K copies, one batched launch vs the dedicated kernel on the same bytes):")
    for s, mult in sorted(distinct.items(), key=lambda kv: -kv[1] * torch.Size(kv[0]).numel()):
        n = torch.Size(s).numel()
        k = max(1, min(32, (64 << 20) // (n * 4)))
        probe_shape(s, k)
    for s in ((4096, 4096), (4096, 11008), (11008, 4096), (1024, 1024), (4096, 1000), (768, 3072), (3072, 768)):
        n = torch.Size(s).numel()
        probe_shape(s, max(1, min(32, (256 << 20) // (n * 4))))
    whole_model("ResNet-50 warm", shapes, 1)
    whole_model("ResNet-50 cold", shapes, 4)
    whole_model("16 x Linear(4096,4096) cold", [(4096, 4096)] * 16, 2, reps=30)
