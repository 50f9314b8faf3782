#!/usr/bin/env python3
"""GPU probe for mctq_qlinear_i8: exact check against an int64 numpy product, then timings per launch variant."""
import sys, os, json, time
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


def oracle(a, za, sa, w, ws, bias):
    acc = (a.astype(np.int64) - za) @ w.astype(np.int64).T
    sc = (np.float32(sa) * ws.astype(np.float32)).astype(np.float32)
    y = (acc.astype(np.int32).astype(np.float32) * sc[None, :]).astype(np.float32)
    return y if bias is None else (y + bias[None, :]).astype(np.float32)


rng = np.random.default_rng(1)
bad = 0
for variant in (0, 41, 42, 44, 81, 82, 84, 181, 182, 184, 142, 144, 1212, 612, 66, 662, 12122):
    lib.mctq_set_tuning(b"ql_variant", variant)
    for (M, N, K) in [(129, 130, 144), (300, 257, 1040), (1, 16, 16), (5, 100, 256), (16, 33, 272), (33, 64, 4096), (64, 4096, 1024), (100, 48, 11008), (7, 1000, 4112)]:
        for u8 in (False, True):
            a = rng.integers(0, 256, (M, K)).astype(np.uint8) if u8 else rng.integers(-128, 128, (M, K)).astype(np.int8)
            w = rng.integers(-128, 128, (N, K)).astype(np.int8)
            za = int(rng.integers(0, 256)) if u8 else int(rng.integers(-128, 128))
            sa = float(rng.uniform(0.001, 0.1))
            ws = rng.uniform(0.001, 0.1, N).astype(np.float32)
            bias = rng.standard_normal(N).astype(np.float32) if (M + N) % 2 else None
            wsum = w.astype(np.int64).sum(1).astype(np.int32)
            y = run(torch.from_numpy(a).to(dev), u8, za, sa, torch.from_numpy(w).to(dev), torch.from_numpy(ws).to(dev),
                    torch.from_numpy(wsum).to(dev), None if bias is None else torch.from_numpy(bias).to(dev), M, N, K)
            want = oracle(a, za, sa, w, ws, bias)
            got = y.cpu().numpy()
            if not np.array_equal(got.view(np.uint32), want.view(np.uint32)):
                bad += 1
                print("MISMATCH", variant, M, N, K, u8, int((got != want).sum()), got.ravel()[:4], want.ravel()[:4])
print("exactness failures:", bad)

# timings: cold weights (ring larger than the 256 MiB Infinity Cache)
res = {}
for (M, N, K) in [(1, 4096, 4096), (16, 4096, 4096), (64, 4096, 4096), (32, 4096, 4096), (16, 11008, 4096), (16, 4096, 11008), (64, 11008, 4096), (64, 4096, 11008), (128, 4096, 4096), (128, 11008, 4096), (256, 4096, 4096)]:
    ring = max(2, int(np.ceil(400e6 / (N * K))))
    ws_ = [torch.randint(-128, 128, (N, K), dtype=torch.int8, device=dev) for _ in range(ring)]
    a = torch.randint(0, 256, (M, K), dtype=torch.uint8, device=dev)
    sc = torch.rand(N, device=dev) * 0.01
    wsum = ws_[0].sum(1, dtype=torch.int32)
    bias = torch.randn(N, device=dev)
    for variant in ((0, 81, 82, 84, 181, 182, 184, 142, 144) if M <= 64 else ((0, 84, 184, 144, 1212, 612, 66, 662, 12122) if M <= 256 else (0, 1212, 612, 66, 662, 12122))):
        lib.mctq_set_tuning(b"ql_variant", variant)
        for i in range(ring + 5):
            run(a, True, 114, 0.02, ws_[i % ring], sc, wsum, bias, M, N, K)
        torch.cuda.synchronize()
        iters = 50
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        y = torch.empty(M, N, dtype=torch.float32, device=dev)
        e0.record()
        for i in range(iters):
            lib.mctq_qlinear_i8(a.data_ptr(), native.CODE_U8, 114, 0.02, ws_[i % ring].data_ptr(), sc.data_ptr(), wsum.data_ptr(),
                                bias.data_ptr(), y.data_ptr(), M, N, K, S())
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1000 / iters
        res[f"M{M}_N{N}_K{K}_v{variant}"] = dict(us=round(us, 2), w_gbs=round(N * K / us / 1e3, 1), tops=round(2 * M * N * K / us / 1e6, 2))
        print(M, N, K, variant, res[f"M{M}_N{N}_K{K}_v{variant}"], flush=True)
    # fp32 reference path on the same shape: torch fp32 linear only (without the fake-quant kernels)
    xf = torch.randn(M, K, device=dev); wf = [torch.randn(N, K, device=dev) for _ in range(max(2, int(np.ceil(400e6 / (4 * N * K)))))]
    for i in range(3): torch.nn.functional.linear(xf, wf[i % len(wf)], bias)
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
    for i in range(20): torch.nn.functional.linear(xf, wf[i % len(wf)], bias)
    e1.record(); torch.cuda.synchronize()
    res[f"M{M}_N{N}_K{K}_torch_fp32_linear"] = dict(us=round(e0.elapsed_time(e1) * 1000 / 20, 2))
    print(M, N, K, "torch fp32 linear", res[f"M{M}_N{N}_K{K}_torch_fp32_linear"], flush=True)
    del ws_, wf
lib.mctq_set_tuning(b"ql_variant", 0)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/qlinear_probe.json", "w"), indent=1)

# 4-bit weights (consumer layout), streaming kernel: time against the int8 path on the same shapes
from mct_quantizers_amd import consumers
for (M, N, K) in [(1, 4096, 4096), (16, 4096, 4096), (32, 4096, 4096), (64, 4096, 4096), (16, 11008, 4096), (16, 4096, 11008), (64, 11008, 4096)]:
    ring = max(2, int(np.ceil(400e6 / (N * K // 2))))
    w4 = [torch.randint(0, 256, (N, K // 2), dtype=torch.uint8, device=dev) for _ in range(ring)]
    a = torch.randint(0, 256, (M, K), dtype=torch.uint8, device=dev)
    sc = torch.rand(N, device=dev) * 0.01
    wsum = torch.zeros(N, dtype=torch.int32, device=dev)
    bias = torch.randn(N, device=dev)
    y = torch.empty(M, N, dtype=torch.float32, device=dev)
    for i in range(5):
        consumers.qlinear_w4a8(a, 114, 0.02, w4[i % ring], sc, wsum, bias)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(50):
        lib.mctq_qlinear_w4a8(a.data_ptr(), native.CODE_U8, 114, 0.02, w4[i % ring].data_ptr(), sc.data_ptr(), wsum.data_ptr(),
                              bias.data_ptr(), y.data_ptr(), -1, 1.0, 0, 0, 0, M, N, K, S())
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / 50
    print(M, N, K, "w4a8", dict(us=round(us, 2), w_gbs=round(N * K / 2 / us / 1e3, 1)), flush=True)
    del w4
