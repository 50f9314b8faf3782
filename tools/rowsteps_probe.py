"""Short whole-step rows (float32 rows of 1024 / 2048 / 3072 elements, 16-bit rows of 2048 / 4096 / 6144): rows_kernel's one- /
two-step tiles against rowsteps_kernel's four steps per block.  Cold (ring larger than the Infinity Cache), HIP events."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mct_quantizers_amd as mq
from mct_quantizers_amd.hip import native
Q = mq.pytorch_quantizers
def timeit(q, xs, steps=400):
    # bench.py's protocol: a fixed-duration run of the same loop first (sustained clocks), outputs of the last launches alive
    keep = [None] * len(xs)
    t0 = time.perf_counter(); i = 0
    while time.perf_counter() - t0 < 0.6:
        keep[i % len(xs)] = q(xs[i % len(xs)]); i += 1
        if i % 128 == 0: torch.cuda.synchronize()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(steps): keep[i % len(xs)] = q(xs[i % len(xs)])
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / steps
t0 = time.perf_counter()
w = torch.randn(4096, 4096, device="cuda"); qw = Q.WeightsSymmetricInferableQuantizer(8, [1.0] * 4096, True, 0)
while time.perf_counter() - t0 < 1.0:
    for _ in range(20): qw(w)
    torch.cuda.synchronize()
for dt, name in ((torch.float32, "f32"), (torch.bfloat16, "bf16"), (torch.float16, "f16")):
    for rows, cols in ((4096, 4096), (8192, 2048), (8192, 3072), (4096, 2048), (2048, 4096), (16384, 4096), (1024, 4096), (11008, 4096), (4096, 6144), (32768, 2048)):
        nbytes = rows * cols * 2 * (4 if dt is torch.float32 else 2)
        ring = max(2, (600 << 20) // nbytes + 1)
        xs = [torch.randn(rows, cols, device="cuda").to(dt) for _ in range(min(ring, 24))]
        q = Q.WeightsSymmetricInferableQuantizer(8, [1.0 + 0.001 * i for i in range(rows)], True, 0)
        out = []
        for rs in (0, 1):
            native.set_tuning("rowsteps", rs)
            us = timeit(q, xs)
            out.append(f"rowsteps={rs}: {us:7.2f} us {nbytes / us / 1e3:6.0f} GB/s ({nbytes / us / 1e3 / 8000:.3f}) {native.last_launch()}")
        print(f"{name} {rows}x{cols} ({nbytes >> 20} MiB per launch, ring {len(xs)}): " + " | ".join(out), flush=True)
        del xs
native.set_tuning("rowsteps", 2)
