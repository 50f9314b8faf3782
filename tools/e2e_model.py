"""End-to-end: a stack of wrapped Linear layers + activation holders, weights re-quantized every forward
(the reference's default behaviour), this package vs the ATen operators the reference would call on the GPU."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mct_quantizers_amd as mq
Q = mq.pytorch_quantizers

class AtenWeights(mq.BaseInferableQuantizer):          # what the reference's WeightsSymmetric.__call__ does on a GPU tensor
    def __init__(self, q): super().__init__(); self.q = q
    def __call__(self, w):
        return torch.fake_quantize_per_channel_affine(w, self.q.scales, self.q.zero_points, 0, -128, 127)
class AtenAct(mq.BaseInferableQuantizer):
    def __init__(self, q): super().__init__(); self.q = q
    def __call__(self, x):
        with torch.no_grad():
            return torch.fake_quantize_per_tensor_affine(x, self.q.scale, self.q.zero_point, 0, 255)

def build(layers, d, aten):
    mods = []
    torch.manual_seed(0)
    for _ in range(layers):
        lin = torch.nn.Linear(d, d, bias=False).cuda()
        thr = [float(v) for v in lin.weight.detach().abs().amax(dim=1)]
        wq = Q.WeightsSymmetricInferableQuantizer(8, thr, True, 0)
        aq = Q.ActivationUniformInferableQuantizer(8, [-3.0], [3.0])
        mods.append(mq.PytorchQuantizationWrapper(lin, {"weight": AtenWeights(wq) if aten else wq}))
        mods.append(mq.PytorchActivationQuantizationHolder(AtenAct(aq) if aten else aq))
    return torch.nn.Sequential(*mods)

def timeit(m, x, n=30):
    with torch.no_grad():
        for _ in range(5): y = m(x)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(n): y = m(x)
        torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3, y

for layers, d, batch in ((16, 4096, 64), (16, 4096, 256), (16, 4096, 2048), (8, 8192, 64)):
    x = torch.randn(batch, d, device="cuda")
    ours, y1 = timeit(build(layers, d, False), x)
    aten, y2 = timeit(build(layers, d, True), x)
    m3 = build(layers, d, False)
    for mod in m3:
        if isinstance(mod, mq.PytorchQuantizationWrapper):
            for q in mod.weights_quantizers.values():
                q.enable_versioned_reuse()
    reuse, y3 = timeit(m3, x)
    # all wrapped weights re-quantized in ONE launch per forward (pytorch/batching.py -> mctq_fq_batched)
    from mct_quantizers_amd.pytorch.batching import batch_weight_quantization
    m6 = build(layers, d, False)
    batch_weight_quantization(m6)
    batched, y6 = timeit(m6, x)
    m7 = build(layers, d, False)
    batch_weight_quantization(m7, reuse_buffers=True)
    planned, y7 = timeit(m7, x)
    # integer consumers: every (activation holder, wrapped Linear) pair runs on the codes (mctq_qlinear_i8)
    from mct_quantizers_amd import consumers
    m4 = build(layers, d, False)
    nf = consumers.fuse_linear_consumers(m4)
    fused, y4 = timeit(m4, x)
    # later layers amplify single quantization-step flips; compare where the two paths first differ: after the
    # first fused pair (wrapped Linear 0, holder 0 / QuantizedLinear 1)
    with torch.no_grad():
        r3 = build(layers, d, False)[:3](x)
        f3 = m4[:3](x)
    err = float((f3 - r3).abs().max() / r3.abs().max())
    m5 = build(layers, d, False)
    consumers.fuse_linear_consumers(m5, chain=True)
    # build()'s order is wrapper, holder, wrapper, ...: every QuantizedLinear but the last feeds the next one
    chained, y5 = timeit(m5, x)
    print(f"{layers} x Linear({d},{d}) batch {batch}: this package {ours:7.3f} ms/forward, ATen fake-quant ops {aten:7.3f} ms/forward "
          f"(x{aten/ours:.2f}), weights batched into one launch {batched:7.3f} ms (pre-packed plan + persistent outputs {planned:7.3f} ms), with versioned weight reuse {reuse:7.3f} ms; "
          f"outputs equal={torch.equal(y1, y2) and torch.equal(y1, y3) and torch.equal(y1, y6) and torch.equal(y1, y7)}; "
          f"{nf} layer pairs on integer codes {fused:7.3f} ms (x{aten/fused:.2f} vs ATen path, max rel diff after the first pair {err:.1e}); "
          f"chained (codes passed between layers) {chained:7.3f} ms (x{aten/chained:.2f}), equal to unchained={torch.equal(y4, y5)}",
          flush=True)


# A convolutional stack (ResNet-ish bottlenecks at 14x14, batch 1): many small weights, so the per-layer quantizer calls are
# launch-bound -- the case batching is for.
def conv_stack(aten):
    torch.manual_seed(1)
    mods = []
    cin = 256
    for i in range(12):
        for cout, k in ((64, 1), (64, 3), (256, 1)):
            conv = torch.nn.Conv2d(cin, cout, k, padding=k // 2, bias=False).cuda()
            thr = [float(v) + 1e-6 for v in conv.weight.detach().abs().amax(dim=(1, 2, 3))]
            wq = Q.WeightsSymmetricInferableQuantizer(8, thr, True, 0)
            aq = Q.ActivationUniformInferableQuantizer(8, [-3.0], [3.0])
            mods.append(mq.PytorchQuantizationWrapper(conv, {"weight": AtenWeights(wq) if aten else wq}))
            mods.append(mq.PytorchActivationQuantizationHolder(AtenAct(aq) if aten else aq))
            cin = cout
    return torch.nn.Sequential(*mods)

x = torch.randn(1, 256, 14, 14, device="cuda")
ours, y1 = timeit(conv_stack(False), x, 50)
aten, y2 = timeit(conv_stack(True), x, 50)
mb = conv_stack(False); batch_weight_quantization(mb); batched, y3 = timeit(mb, x, 50)
mp = conv_stack(False); batch_weight_quantization(mp, reuse_buffers=True); planned, y4 = timeit(mp, x, 50)
# the same forward replayed from one hipGraph (every kernel of the package is capture-legal): host cost gone
def graphed_ms(model):
    xs_ = x.clone()
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side), torch.no_grad():
        for _ in range(3): model(xs_)
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g), torch.no_grad():
        yg = model(xs_)
    for _ in range(5): g.replay()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(50): g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / 50 * 1e3, yg
g1, yg1 = graphed_ms(conv_stack(False))
mgb = conv_stack(False); batch_weight_quantization(mgb, reuse_buffers=True)
g2, yg2 = graphed_ms(mgb)
print(f"  ... replayed from one hipGraph: this package {g1:7.3f} ms/forward, with the weights batched into one launch {g2:7.3f} ms; "
      f"equal={torch.equal(yg1, y1) and torch.equal(yg2, y1)}", flush=True)
try:
    g3, yg3 = graphed_ms(conv_stack(True))
    print(f"  ... ATen fake-quant ops under the same capture: {g3:7.3f} ms", flush=True)
except Exception as e:  # noqa: BLE001
    print(f"  ... ATen's fake_quantize_per_channel_affine cannot be captured (its zero-point range check reads the device): {str(e).splitlines()[0]}", flush=True)
    torch.cuda.synchronize()
print(f"36 wrapped convolutions (1x1 / 3x3, 64-256 channels) + holders, batch 1 at 14x14: this package {ours:7.3f} ms/forward, "
      f"ATen fake-quant ops {aten:7.3f} ms (x{aten/ours:.2f}), weights batched {batched:7.3f} ms, pre-packed plan {planned:7.3f} ms "
      f"(x{aten/planned:.2f}); outputs equal={torch.equal(y1, y2) and torch.equal(y1, y3) and torch.equal(y1, y4)}", flush=True)
