"""Short rows (inner < 1024 elements): the window kernel (parameters of a tile's rows staged in LDS) against the batched
kernel's per-lane-vector table reads, same tensors, cold ring."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mct_quantizers_amd as mq
from mct_quantizers_amd.hip import native, ops
Q = mq.pytorch_quantizers
def timeit(f, xs, steps=100):
    for i in range(len(xs) + 5): f(xs[i % len(xs)])
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(steps): f(xs[i % len(xs)])
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / steps
for dt in (torch.float32, torch.bfloat16):
    for shape, axis in (((65536, 256), 0), ((262144, 64), 0), ((1048576, 16), 0), ((16384, 1020), 0), ((4096, 576), 0), ((2048, 2304), 0), ((64, 256, 56, 56), 1), ((512, 512, 3, 3), 0)):
        C = shape[axis]
        q = Q.WeightsSymmetricInferableQuantizer(8, [1.0 + 0.001 * (i % 997) for i in range(C)], True, axis)
        n = int(np.prod(shape)); nbytes = n * 2 * (4 if dt == torch.float32 else 2)
        ring = max(2, -(-(512 << 20) // nbytes) + 1)
        xs = [torch.randn(shape, device="cuda").to(dt) for _ in range(min(ring, 40))]
        a = timeit(q, xs); va = native.last_launch()
        b = timeit(lambda t: ops.fq_batched([q.batch_item(t)])[0], xs); vb = native.last_launch()
        same = torch.equal(q(xs[0]), ops.fq_batched([q.batch_item(xs[0])])[0])
        print(f"{str(dt):15s} {str(shape):20s} axis {axis}: {va.split('<')[0]:16s} {a:7.2f} us {nbytes/a/1e3:6.0f} GB/s | {vb.split('<')[0]:16s} {b:7.2f} us {nbytes/b/1e3:6.0f} GB/s  equal={same}", flush=True)
