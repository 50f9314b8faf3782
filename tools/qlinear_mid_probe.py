#!/usr/bin/env python3
"""mctq_qlinear_i8 between its two regimes (128 < M < 2048) on 4096 x 4096 weights: time per launch variant and per
tile order (tuning keys ql_variant / ql_band), cold weights (ring > Infinity Cache), every result compared bit for bit
with the exact integer product (float64 matmul of the codes is exact here: |sum| < 2^53) and the float32 epilogue."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from mct_quantizers_amd.hip import native

lib = native.load()
dev = torch.device("cuda")
S = lambda: torch.cuda.current_stream().cuda_stream
def _shape(a):                                   # "M" (4096 x 4096 weights) or "MxNxK"
    p = [int(v) for v in a.split("x")]
    return (p[0], 4096, 4096) if len(p) == 1 else tuple(p)
shapes = [_shape(a) for a in (sys.argv[1].split(",") if len(sys.argv) > 1 else "256,512,1024,2048".split(","))]
variants = [int(v) for v in (sys.argv[2].split(",") if len(sys.argv) > 2 else "0,662,66,612,1212,12122,2548,2560".split(","))]
bands = [int(v) for v in (sys.argv[3].split(",") if len(sys.argv) > 3 else "0,1".split(","))]
PEAK = 5000.0
if os.environ.get("QL_STAGGER"):                  # half of the tiled kernel's waves copy after multiplying (default on)
    assert lib.mctq_set_tuning(b"ql_stagger", int(os.environ["QL_STAGGER"])) == 0
if os.environ.get("QL_ROT"):                      # K rotation between the blocks that share a weight tile (default on)
    assert lib.mctq_set_tuning(b"ql_rot", int(os.environ["QL_ROT"])) == 0
for (M, N, K) in shapes:
    ring = int(os.environ["QL_RING"]) if os.environ.get("QL_RING") else max(2, int(np.ceil(400e6 / (N * K))))   # 1: warm weights
    ws = [torch.randint(-128, 128, (N, K), dtype=torch.int8, device=dev) for _ in range(ring)]
    a = torch.randint(0, 256, (M, K), dtype=torch.uint8, device=dev)
    sc = torch.rand(N, device=dev) * 0.01
    wsum = ws[0].sum(1, dtype=torch.int32)
    bias = torch.randn(N, device=dev)
    za, sa = 131, 0.02
    acc = ((a.double() - za) @ ws[0].double().T)                          # exact
    want = (acc.float() * (torch.tensor(sa, dtype=torch.float32, device=dev) * sc)[None, :]) + bias[None, :]
    ys = [torch.empty(M, N, dtype=torch.float32, device=dev) for _ in range(ring)]
    for v in variants:
        for b in (bands if v in (0, 662, 66, 612, 1212, 12122) else [0]):
            assert lib.mctq_set_tuning(b"ql_variant", v) == 0 and lib.mctq_set_tuning(b"ql_band", b) == 0

            def call(i):
                return lib.mctq_qlinear_i8(a.data_ptr(), native.CODE_U8, za, sa, ws[i % ring].data_ptr(), sc.data_ptr(), wsum.data_ptr(),
                                           bias.data_ptr(), ys[i % ring].data_ptr(), M, N, K, S())
            rc = call(0)
            if rc != 0:
                print(f"M={M:5d} N={N} K={K} variant {v:6d} band {b}: not applicable ({lib.mctq_last_error().decode()[:60]})")
                continue
            torch.cuda.synchronize()
            ok = bool(torch.equal(ys[0], want))
            for i in range(5): call(i)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 60
            e0.record()
            for i in range(reps): call(i)
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / reps
            tops = 2.0 * M * N * K / us / 1e6
            print(f"M={M:5d} N={N} K={K} variant {v:6d} band {b}: {us:8.2f} us  {tops:7.1f} TOP/s  {tops / PEAK:.3f} of int8 peak  exact={ok}", flush=True)
lib.mctq_set_tuning(b"ql_variant", 0); lib.mctq_set_tuning(b"ql_band", 0)
