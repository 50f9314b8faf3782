"""Launch-shape probe on the config-4 tensor (4096 x 11008 float32, rows of 2752 lane-vectors = 2.69 tiles of 1024):
the LUT kernel with each heavy_unroll / persistent setting, and -- same tensor, AFFINE op -- the per-row tile launch
(rows_kernel) against tiles that ignore row boundaries (the batched kernel's shape)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mct_quantizers_amd as mq
from mct_quantizers_amd import workloads
from mct_quantizers_amd.hip import native, ops
Q = mq.pytorch_quantizers
x_np = workloads.make_input("cfg4"); wl = workloads.make_workload("cfg4", x_np)
ring = 3
xs = [torch.from_numpy(x_np).cuda() for _ in range(ring)]
def timeit(f, steps=100):
    for i in range(10): f(xs[i % ring])
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(steps): y = f(xs[i % ring])
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / steps
nbytes = x_np.size * 8
qlut = getattr(Q, wl.quantizer)(**wl.kwargs)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 1.0:          # clocks up before the first variant is timed
    for i in range(20): qlut(xs[i % ring])
    torch.cuda.synchronize()
for hu in (0, 1, 2, 4):
    native.set_tuning("heavy_unroll", hu)
    us = timeit(qlut); print(f"LUT table kernel heavy_unroll={hu}: {us:7.2f} us {nbytes/us/1e3:6.0f} GB/s  {native.last_launch()}", flush=True)
native.set_tuning("heavy_unroll", 0)
thr = wl.kwargs["threshold"]
qa = Q.WeightsSymmetricInferableQuantizer(8, thr, True, 0)
for u in (1, 2, 4):
    native.set_tuning("unroll", u)
    us = timeit(qa); print(f"affine per-row tiles unroll<={u}: {us:7.2f} us {nbytes/us/1e3:6.0f} GB/s  {native.last_launch()}", flush=True)
native.set_tuning("unroll", 4)
us = timeit(lambda t: ops.fq_batched([qa.batch_item(t)])[0]); print(f"affine tiles across row boundaries (batched kernel): {us:7.2f} us {nbytes/us/1e3:6.0f} GB/s  {native.last_launch()}", flush=True)
x2 = [t[:, :10240].contiguous() for t in xs]; xs = x2; nbytes = xs[0].numel() * 8
qa2 = Q.WeightsSymmetricInferableQuantizer(8, thr, True, 0)
us = timeit(qa2); print(f"affine, rows of exactly 2.5 tiles (4096 x 10240): {us:7.2f} us {nbytes/us/1e3:6.0f} GB/s  {native.last_launch()}", flush=True)
x2 = [t[:, :8192].contiguous() for t in xs]; xs = x2; nbytes = xs[0].numel() * 8
us = timeit(qa2); print(f"affine, rows of exactly 2 tiles (4096 x 8192): {us:7.2f} us {nbytes/us/1e3:6.0f} GB/s  {native.last_launch()}", flush=True)
