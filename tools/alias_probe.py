#!/usr/bin/env python3
"""Does the relative placement of input and output matter?  cfg2 (4096 x 4096 float32, per channel) through the raw C
ABI with x_i and y_i carved out of one big allocation: y_i = x_i + 64 MiB + delta, ring of 5 pairs (cold protocol).
Prints the launch period for each delta."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mct_quantizers_amd.hip import native

lib = native.load()
dev = torch.device("cuda")
C = inner = 4096
nbytes = C * inner * 4
big = torch.empty(5 * 256 * (1 << 20) + (64 << 20), dtype=torch.uint8, device=dev)
base = (big.data_ptr() + (1 << 21) - 1) & ~((1 << 21) - 1)          # 2 MiB aligned
src = torch.randn(C, inner, device=dev)
scales = torch.rand(C, device=dev) * 0.01 + 0.001
zps = torch.zeros(C, dtype=torch.int32, device=dev)
S = torch.cuda.current_stream().cuda_stream
import ctypes
hip = ctypes.CDLL("libamdhip64.so")
print("base mod 1GiB:", hex(base & ((1 << 30) - 1)))
ring = 5
stride = 256 << 20
for i in range(ring):
    hip.hipMemcpyAsync(ctypes.c_void_p(base + i * stride), ctypes.c_void_p(src.data_ptr()), ctypes.c_size_t(nbytes), 3, ctypes.c_void_p(S))
torch.cuda.synchronize()


def period(delta, gap=64 << 20, n=600):
    def call(i):
        x = base + (i % ring) * stride
        return lib.mctq_fq_per_channel_f32(x, x + gap + delta, 1, C, inner, scales.data_ptr(), zps.data_ptr(), -128, 127, S)
    for i in range(50):
        assert call(i) == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        call(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


# pre-warm clocks
period(0, n=20000)
for rep in range(2):
    for delta in (0, 256, 1024, 4096, 8192, 16384, 32768, 65536, 131072, 262144, 524288, 1 << 20, 3 << 19, 2 << 20, 5 << 20):
        us = period(delta)
        print(f"rep {rep} delta {delta:8d} B: {us:6.2f} us  {2 * nbytes / us / 1e6:5.2f} TB/s  frac {2 * nbytes / us / 1e6 / 8:.3f}", flush=True)
