#!/usr/bin/env python3
"""Channel-last per-channel shapes (lastaxis_kernel) under bench.py's protocol (0.5 s pre-warm per shape, cold ring, outputs kept,
HIP events around 200 launches): one line per (storage type, shape).  Run once per library build to compare rows-per-lane
choices (MCTQ_HIP_LIB=tools/ablate/libmctq_hip_lu_X.so MCTQ_BINDING=ctypes)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mct_quantizers_amd as mq
from mct_quantizers_amd.hip import native
Q = mq.pytorch_quantizers
tag = os.environ.get("MCTQ_HIP_LIB", "shipped").split("_")[-1].replace(".so", "")
for dt in (torch.bfloat16, torch.float16, torch.float32):
    for shape in ((4096, 4096), (2048, 8192), (8192, 2048), (1024, 4096), (8192, 8192), (64, 56, 56, 256), (16, 112, 112, 64),
                  (1048576, 16), (50257, 768), (256, 14, 14, 1024)):
        C = shape[-1]
        x = (torch.randn(*shape, device="cuda") * 2).to(dt)
        nb = x.numel() * x.element_size() * 2
        ring = min(64, max(2, -(-(512 << 20) // nb) + 1))
        xs = [x] + [x.clone() for _ in range(ring - 1)]
        ys = [None] * ring
        q = Q.WeightsSymmetricInferableQuantizer(8, [1.0 + (i % 97) * 0.01 for i in range(C)], True, len(shape) - 1)
        def call(i):
            ys[i % ring] = q(xs[i % ring])
        t0, n = time.perf_counter(), 0
        while time.perf_counter() - t0 < 0.5:
            call(n); n += 1
            if n % 256 == 0: torch.cuda.synchronize()
        res = []
        for _ in range(3):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(200): call(i)
            e1.record(); torch.cuda.synchronize()
            res.append(e0.elapsed_time(e1) * 1e3 / 200)
        res.sort()
        print(f"{tag:8s} {str(dt)[6:]:9s} {str(shape):20s} {res[0]:8.2f} {res[1]:8.2f} us  {nb / res[1] / 1e3:6.0f} GB/s  {native.last_launch()}", flush=True)
        del xs, ys
        torch.cuda.empty_cache()
