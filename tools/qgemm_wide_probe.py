#!/usr/bin/env python3
"""GPU probe for the dense int8 consumer (mctq_qlinear_i8 with many rows): exact check of the wide-tile kernels against a
float64 product (exact for K <= 32768), then timings per tile variant on the dense shapes of tools/bench_consumer.py."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from mct_quantizers_amd.hip import native

lib = native.load()
dev = torch.device("cuda")
S = lambda: torch.cuda.current_stream().cuda_stream


def run(a, u8, za, sa, w, ws, wsum, bias, M, N, K):
    y = torch.empty(M, N, dtype=torch.float32, device=dev)
    rc = lib.mctq_qlinear_i8(a.data_ptr(), native.CODE_U8 if u8 else native.CODE_I8, za, sa, w.data_ptr(), ws.data_ptr(),
                             wsum.data_ptr(), bias.data_ptr() if bias is not None else None, y.data_ptr(), M, N, K, S())
    assert rc == 0, lib.mctq_last_error()
    return y


def want(a, za, sa, w, ws, bias):
    acc = (a.double() - za) @ w.double().T
    y = acc.to(torch.int32).float() * (torch.tensor(sa, dtype=torch.float32, device=dev) * ws)
    return y if bias is None else y + bias


variants = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [2588, 2548, 2584]
g = torch.Generator(device=dev).manual_seed(1)
bad = 0
for variant in variants:
    lib.mctq_set_tuning(b"ql_variant", variant)
    for (M, N, K) in [(256, 256, 256), (256, 256, 384), (512, 768, 640), (768, 512, 512), (1024, 1024, 4096), (256, 512, 11008), (128, 256, 512), (384, 512, 320)]:
        for u8 in (False, True):
            a = torch.randint(0, 256, (M, K), dtype=torch.uint8, device=dev, generator=g) if u8 else \
                torch.randint(-128, 128, (M, K), dtype=torch.int8, device=dev, generator=g)
            w = torch.randint(-128, 128, (N, K), dtype=torch.int8, device=dev, generator=g)
            za = 77 if u8 else -5
            sa = 0.0173
            ws = torch.rand(N, device=dev, generator=g) * 0.05 + 0.001
            bias = torch.randn(N, device=dev, generator=g) if (M // 256 + N // 256) % 2 else None
            wsum = w.sum(1, dtype=torch.int32)
            try:
                y = run(a, u8, za, sa, w, ws, wsum, bias, M, N, K)
            except AssertionError:
                continue                                   # this tile shape does not take the problem: refused, fine
            ref = want(a, za, sa, w, ws, bias)
            if not torch.equal(y.view(torch.int32), ref.view(torch.int32)):
                bad += 1
                print("MISMATCH", variant, M, N, K, u8, int((y != ref).sum()), y.ravel()[:4].tolist(), ref.ravel()[:4].tolist())
print("exactness failures:", bad, flush=True)

for (M, N, K) in [(256, 4096, 4096), (512, 4096, 4096), (1024, 4096, 4096), (2048, 4096, 4096), (4096, 4096, 4096), (4096, 4096, 11008), (4096, 11008, 4096), (8192, 8192, 8192)]:
    a = torch.randint(0, 256, (M, K), dtype=torch.uint8, device=dev)
    w = torch.randint(-128, 128, (N, K), dtype=torch.int8, device=dev)
    sc = torch.rand(N, device=dev) * 0.01
    wsum = w.sum(1, dtype=torch.int32)
    bias = torch.randn(N, device=dev)
    base = None
    for variant in [0, 1212] + variants:
        lib.mctq_set_tuning(b"ql_variant", variant)
        try:
            y = run(a, True, 3, 0.02, w, sc, wsum, bias, M, N, K)
        except AssertionError as e:
            print(f"M={M} N={N} K={K} variant {variant}: {e}")
            continue
        if base is None:
            base = y
        eq = torch.equal(y, base)
        for _ in range(3):
            run(a, True, 3, 0.02, w, sc, wsum, bias, M, N, K)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20
        e0.record()
        for _ in range(n):
            run(a, True, 3, 0.02, w, sc, wsum, bias, M, N, K)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / n * 1e3
        print(f"M={M} N={N} K={K} variant {variant:6d}: {us:9.1f} us  {2.0 * M * N * K / us / 1e9:8.1f} TOP/s  equal={eq}", flush=True)
lib.mctq_set_tuning(b"ql_variant", 0)
