"""Short / ragged rows: the single-tensor per-channel entry point (window / lastaxis kernels) vs the SAME tensor as a
one-item batched launch (per-lane-vector gather, no LDS window, no block barrier).  Cold ring, event-timed, bit-compared."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mct_quantizers_amd.hip import native, ops

fast = native.fast()
for dt in (torch.float32, torch.bfloat16):
    for shape, axis in (((16384, 1020), 0), ((65536, 256), 0), ((262144, 64), 0), ((1048576, 16), 0), ((4096, 4100), 0),
                        ((4096, 4099), 0), ((50257, 768), 0), ((8192, 2056), 1), ((65536, 200), 1), ((64, 256, 56, 56), 1),
                        ((4096, 4096), 0)):
        ring = 4
        xs = [(torch.randn(shape, device="cuda") * 2).to(dt) for _ in range(ring)]
        c = shape[axis]
        s = torch.rand(c, device="cuda") * 0.05 + 0.01
        z = torch.zeros(c, dtype=torch.int32, device="cuda")
        ys = [torch.empty_like(x) for x in xs]
        plans = [fast.BatchPlan([(x, y, s, None, axis, -128, 127)]) for x, y in zip(xs, ys)]
        single = lambda i: ops.fq_per_channel(xs[i % ring], s, z, axis, -128, 127, True)
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
        nb = xs[0].numel() * xs[0].element_size() * 2
        print(f"{str(dt)[6:]:9s}{str(shape):22s} axis {axis}  single {res['single'][0]:7.2f} us {nb / res['single'][0] / 1e3:6.0f} GB/s [{res['single'][1].split('<')[0]}]"
              f"   one-item batched {res['batched-1'][0]:7.2f} us {nb / res['batched-1'][0] / 1e3:6.0f} GB/s  equal={same}", flush=True)
