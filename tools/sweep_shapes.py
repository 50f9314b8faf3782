"""Throughput sweep of the per-channel / per-tensor affine kernels over shapes, axes and storage types (cold ring), to spot launch
shapes that fall off the HBM roof.

Protocol (round 6): per shape a ring of input buffers (> 512 MiB of traffic between two uses of a buffer) AND a
ring of outputs kept alive; warm-up = 0.3 s of the same loop and at least TWO full passes over the ring, so that every output slot
has been allocated before the clock starts (round 5 warmed up with 5 calls over rings of up to 64 buffers: the first timed pass
then paid torch's allocator -- hipMalloc -- for every fresh output, which is what the 18 us readings on 512 x 512 x 3 x 3 were;
`--explain-outlier` reproduces both protocols on that shape and counts the device allocations inside the timed region);
5 repeats of 100 launches between HIP events: min / median / max of the repeat means, the launch variant per line.

    python tools/sweep_shapes.py [--dtypes f32,bf16,f16] [--routes 0,1,2] [--explain-outlier] [--quick]
`--routes`: tuning key "shortrows" values to sweep for 16-bit storage (default: the library's default only)."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mct_quantizers_amd as mq
from mct_quantizers_amd.hip import native
Q = mq.pytorch_quantizers
ap = argparse.ArgumentParser()
ap.add_argument("--dtypes", default="f32,bf16")
ap.add_argument("--routes", default="")
ap.add_argument("--explain-outlier", action="store_true")
ap.add_argument("--quick", action="store_true")
args = ap.parse_args()
DT = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}

SHAPES = (((4096, 4096), 0), ((4096, 4096), 1), ((4096, 4100), 0), ((4096, 4099), 0), ((4097, 4096), 0),
          ((16384, 1024), 0), ((16384, 1020), 0), ((65536, 256), 0), ((262144, 64), 0), ((1048576, 16), 0),
          ((2048, 8192), 1), ((256, 65536), 0), ((8, 2097152), 0), ((1, 16777216), 0),
          ((64, 256, 56, 56), 1), ((64, 56, 56, 256), 3), ((512, 512, 3, 3), 0), ((2048, 2048, 3, 3), 0),
          ((32, 3, 224, 224), 1), ((50257, 768), 0), ((50257, 768), 1), ((1048576, 16), 1), ((65536, 200), 1),
          ((8192, 2056), 1), ((8192, 8192), 1), ((256, 14, 14, 1024), 3), ((4194304, 8), 0), ((131072, 136), 0))
if args.quick:
    SHAPES = SHAPES[:4] + (((16384, 1020), 0), ((1048576, 16), 0), ((512, 512, 3, 3), 0), ((65536, 200), 1))


def measure(f, xs, repeats=5, steps=100, warm_s=0.3):
    n = len(xs)
    outs = [None] * n
    t0, k = time.perf_counter(), 0
    while k < 2 * n or time.perf_counter() - t0 < warm_s:      # every output slot allocated, clocks up
        outs[k % n] = f(xs[k % n]); k += 1
        if k % 256 == 0:
            torch.cuda.synchronize()
    res = []
    for _ in range(repeats):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(steps):
            outs[i % n] = f(xs[i % n])
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) * 1e3 / steps)
    res.sort()
    return res[0], res[len(res) // 2], res[-1]


def ring_for(x):
    b = x.numel() * x.element_size() * 2
    ring = min(64, max(2, -(-(512 << 20) // b) + 1))
    return b, [x] + [x.clone() for _ in range(ring - 1)]


_w = torch.randn(4096, 4096, device="cuda"); _q = Q.WeightsSymmetricInferableQuantizer(8, [1.0] * 4096, True, 0)
for _ in range(3000): _q(_w)          # bring the clocks up before the first measurement
torch.cuda.synchronize()

if args.explain_outlier:
    shape, axis = (512, 512, 3, 3), 0
    q = Q.WeightsSymmetricInferableQuantizer(8, [1.0 + (i % 97) * 0.01 for i in range(512)], True, axis)
    for name in ("f32", "bf16"):
        x = torch.randn(*shape, device="cuda").to(DT[name])
        b, xs = ring_for(x)
        n = len(xs)
        torch.cuda.empty_cache()
        # round 5's protocol: five warm-up calls, then the clock
        outs = [None] * n
        for i in range(5): outs[i % n] = q(xs[i % n])
        torch.cuda.synchronize()
        a0 = torch.cuda.memory_stats()["num_device_alloc"]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(100): outs[i % n] = q(xs[i % n])
        e1.record(); torch.cuda.synchronize()
        a1 = torch.cuda.memory_stats()["num_device_alloc"]
        old = e0.elapsed_time(e1) * 1e3 / 100
        lo, med, hi = measure(q, xs)
        a2 = torch.cuda.memory_stats()["num_device_alloc"]
        print(f"{name} {shape}: ring {n}; round-5 protocol {old:6.2f} us per launch with {a1 - a0} device allocations (hipMalloc) inside "
              f"the timed 100 launches; round-6 protocol {lo:5.2f} / {med:5.2f} / {hi:5.2f} us (min / median / max of 5 x 100), "
              f"{a2 - a1} device allocations from its warm-up on  [{native.last_launch()}]", flush=True)
        del xs, outs
    sys.exit(0)

print("dtype     shape                  axis   min / median / max us   GB/s (median)  frac   spread  launch", flush=True)
for name in args.dtypes.split(","):
    dt = DT[name]
    routes = [int(r) for r in args.routes.split(",")] if args.routes and dt != torch.float32 else [None]
    for shape, axis in SHAPES:
        C = shape[axis]
        q = Q.WeightsSymmetricInferableQuantizer(8, [1.0 + (i % 97) * 0.01 for i in range(C)], True, axis)
        x = torch.randn(*shape, device="cuda").to(dt)
        b, xs = ring_for(x)
        for route in routes:
            if route is not None:
                native.set_tuning("shortrows", route)
            lo, med, hi = measure(q, xs)
            tag = "" if route is None else f" shortrows={route}"
            flag = "  <-- spread > 1.3x" if hi > 1.3 * lo else ""
            print(f"{name:9s} {str(shape):22s} {axis:4d}  {lo:7.2f} /{med:7.2f} /{hi:7.2f}   {b / med / 1e3:7.0f}      {b / med / 8e6:.3f}  "
                  f"{hi / lo:5.2f}x  {native.last_launch()}{tag}{flag}", flush=True)
        if routes != [None]:
            native.set_tuning("shortrows", 1)
        del xs
        torch.cuda.empty_cache()
