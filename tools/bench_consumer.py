#!/usr/bin/env python3
"""One JSON line per problem size for the integer consumer (mctq_qlinear_i8), in bench.py's vocabulary: HIP-event
time per launch over cold weights, and the roofline that bounds it -- streaming the weight codes once (hbm) for few
rows, the dense int8 MFMA peak (2x the bf16 figure of MI355X_MICROARCH.md) for many.  Not the judged bench line.
Every line records the git head it was taken at (argv[1]: the GPU box has no .git) and the launch variant the library
chose (mctq_last_launch), so a row cannot outlive the heuristic that produced it unnoticed."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from mct_quantizers_amd.hip import native

HBM_PEAK_GBS, I8_PEAK_TOPS = 8000.0, 5000.0
GIT_HEAD = sys.argv[1] if len(sys.argv) > 1 else os.environ.get("MCTQ_GIT_HEAD", "unknown")
lib = native.load(); dev = torch.device("cuda"); S = lambda: torch.cuda.current_stream().cuda_stream
for (M, N, K) in [(16, 4096, 4096), (16, 11008, 4096), (32, 4096, 4096), (64, 4096, 4096), (64, 4096, 11008), (128, 4096, 4096), (256, 4096, 4096), (256, 11008, 4096), (512, 4096, 4096), (768, 4096, 4096), (1024, 4096, 4096), (2048, 4096, 4096), (4096, 4096, 4096), (4096, 4096, 11008), (8192, 8192, 8192)]:
    ring = max(2, int(np.ceil(400e6 / (N * K))))
    ws = [torch.randint(-128, 128, (N, K), dtype=torch.int8, device=dev) for _ in range(ring)]
    a = torch.randint(0, 256, (M, K), dtype=torch.uint8, device=dev)
    sc = torch.rand(N, device=dev) * 0.01; rs = ws[0].sum(1, dtype=torch.int32); bias = torch.randn(N, device=dev)
    y = torch.empty(M, N, device=dev)
    call = lambda i: lib.mctq_qlinear_i8(a.data_ptr(), native.CODE_U8, 114, 0.02, ws[i % ring].data_ptr(), sc.data_ptr(),
                                         rs.data_ptr(), bias.data_ptr(), y.data_ptr(), M, N, K, S())
    for i in range(ring + 10): call(i)
    variant = native.last_launch()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    steps = 200; e0.record()
    for i in range(steps): call(i)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / steps
    ops, wbytes = 2.0 * M * N * K, float(N * K + M * K + M * N * 4)
    t_mfma, t_hbm = ops / (I8_PEAK_TOPS * 1e12), wbytes / (HBM_PEAK_GBS * 1e9)
    if t_hbm >= t_mfma:
        roof = {"bound": "hbm", "achieved": wbytes / us / 1e3, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": wbytes / us / 1e3 / HBM_PEAK_GBS}
    else:
        roof = {"bound": "mfma", "achieved": ops / us / 1e6, "peak": I8_PEAK_TOPS, "unit": "TOP/s", "frac": ops / us / 1e6 / I8_PEAK_TOPS}
    print(json.dumps({"metric": "int8 multiply-accumulate ops/s of the integer consumer", "value": ops / (us * 1e-6), "unit": "op/s",
                      "n_gpus": 1, "steps": steps, "ms_per_step": us / 1e3, "dtype": "i8", "data": "synthetic codes",
                      "config": {"workload": f"mctq_qlinear_i8 M={M} N={N} K={K}", "cache_protocol": "cold", "buffer_ring": ring,
                                 "launch_variant": variant, "git_head": GIT_HEAD},
                      "roofline": roof}), flush=True)
    del ws
