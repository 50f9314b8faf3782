"""Time the three output forms of one affine quantizer on cold buffers: fake-quant (float32 out), int8 codes, packed
4-bit codes.  Algorithmic bytes per float32 element: 8, 5 and 4.5."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mct_quantizers_amd as mq
Q = mq.pytorch_quantizers


def timeit(f, xs, steps=200):
    n = len(xs); outs = [None] * n
    for i in range(20): outs[i % n] = f(xs[i % n])
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(steps): outs[i % n] = f(xs[i % n])
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) * 1e3 / steps


_w = torch.randn(4096, 4096, device="cuda"); _q = Q.WeightsSymmetricInferableQuantizer(8, [1.0] * 4096, True, 0)
for _ in range(3000): _q(_w)
for shape, dt in (((8192, 8192), torch.float32), ((4096, 4096), torch.float32), ((8192, 8192), torch.bfloat16)):
    C = shape[0]
    q = Q.WeightsPOTInferableQuantizer(4, [2.0 ** ((i % 5) - 2) for i in range(C)], True, 0)
    x = torch.randn(*shape, device="cuda").to(dt)
    ring = max(3, -(-(600 << 20) // (x.numel() * x.element_size() * 2)))
    xs = [x] + [x.clone() for _ in range(ring - 1)]
    n, eb = x.numel(), x.element_size()
    for name, f, out_b in (("fake-quant", q, eb), ("int8 codes", lambda t: q.quantize_to_codes(t)[0], 1),
                           ("4-bit codes", lambda t: q.quantize_to_codes(t, packed4=True)[0], 0.5)):
        t = timeit(f, xs)
        print(f"{str(dt)[6:]:9s} {str(shape):14s} {name:12s} {t:8.2f} us  {n * (eb + out_b) / t / 1e3:7.0f} GB/s", flush=True)
