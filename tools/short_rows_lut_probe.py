"""LUT quantizers on short / ragged float32 rows: the single-tensor entry point (window_kernel<LutTableOp>) vs the same
tensor as a one-item batched LUT launch (per-lane-vector thresholds, table in LDS).  Cold ring, event-timed, bit-compared."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from mct_quantizers_amd.hip import native, ops

fast = native.fast()
LUT16 = np.float32([-128, -96, -64, -40, -24, -12, -5, 0, 5, 12, 24, 40, 64, 96, 120, 127])
table = ops.make_lut_table(LUT16, 128.0, -128.0, 127.0, "cuda")
lut = torch.from_numpy(LUT16).cuda()
for shape, axis in (((16384, 1020), 0), ((65536, 256), 0), ((262144, 64), 0), ((4096, 4099), 0), ((50257, 768), 0),
                    ((2048, 512, 1, 1), 0), ((4096, 4096), 0), ((4096, 11008), 0)):
    ring = 4
    xs = [torch.randn(shape, device="cuda") for _ in range(ring)]
    c = shape[axis]
    thr = torch.rand(c, device="cuda") * 2 + 1.0
    ys = [torch.empty(shape, dtype=torch.float32, device="cuda") for _ in xs]
    plans = [fast.BatchPlan([("lut", x, y, thr, table, axis, 1e-8, 0.0, 0.0, 128.0, -128.0, 127.0, 0)]) for x, y in zip(xs, ys)]
    single = lambda i: ops.lut_per_channel(xs[i % ring], lut, thr, 1e-8, axis, 128.0, -128.0, 127.0, table)
    batched = lambda i: plans[i % ring]()
    res = {}
    for name, f in (("single", single), ("batched-1", batched)):
        for i in range(8): f(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(100): f(i)
        e1.record(); torch.cuda.synchronize()
        res[name] = (e0.elapsed_time(e1) * 10, native.last_launch())
    same = torch.equal(single(0), ys[0])
    nb = xs[0].numel() * 8
    print(f"{str(shape):22s} axis {axis}  single {res['single'][0]:7.2f} us {nb / res['single'][0] / 1e3:6.0f} GB/s [{res['single'][1].split('<')[0]}]"
          f"   one-item batched {res['batched-1'][0]:7.2f} us {nb / res['batched-1'][0] / 1e3:6.0f} GB/s  equal={same}", flush=True)
