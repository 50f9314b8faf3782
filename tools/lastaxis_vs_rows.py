#!/usr/bin/env python3
"""4096 x 4096 bfloat16 (or float32) per-channel along axis 0 (rows / rowsteps kernel) and along axis 1 (lastaxis kernel) on the
same cold ring: time per launch, and -- under `rocprofv3 --kernel-trace --pmc ...` with argument `pmc` -- 40 launches of each
for the counters.  The channel-last form of 16-bit tensors is the library's slowest large-tensor launch (0.61 of 8 TB/s)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mct_quantizers_amd as mq
from mct_quantizers_amd.hip import native
Q = mq.pytorch_quantizers
pmc = len(sys.argv) > 1 and sys.argv[1] == "pmc"
for dt in (torch.bfloat16, torch.float32):
    x = (torch.randn(4096, 4096, device="cuda") * 2).to(dt)
    ring = 9 if dt is torch.bfloat16 else 5
    xs = [x] + [x.clone() for _ in range(ring - 1)]
    ys = [None] * ring
    for axis in (0, 1):
        q = Q.WeightsSymmetricInferableQuantizer(8, [1.0 + (i % 97) * 0.01 for i in range(4096)], True, axis)
        def call(i):
            ys[i % ring] = q(xs[i % ring])
        for i in range(ring + 3): call(i)
        torch.cuda.synchronize()
        if pmc:
            for i in range(40): call(i)
            torch.cuda.synchronize()
            continue
        t0, n = time.perf_counter(), 0
        while time.perf_counter() - t0 < 0.7:
            call(n); n += 1
            if n % 256 == 0: torch.cuda.synchronize()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(300): call(i)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 300
        nb = x.numel() * x.element_size() * 2
        print(f"{str(dt)[6:]:9s} axis {axis}: {us:7.2f} us  {nb / us / 1e3:6.0f} GB/s ({nb / us / 1e3 / 8000:.3f})  {native.last_launch()}", flush=True)
