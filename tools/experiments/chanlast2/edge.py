#!/usr/bin/env python3
"""Where does shortrows_kernel's edge over the per-tensor kernel come from?  The SAME 4096 x 4096 16-bit buffers, the shipped library, arms
timed alternately (A B C D A B C D ...; bench.py's cold ring, HIP events around 200 launches, 5 rounds each):
  A  per channel along axis 0, 4096 different scales                         (shortrows_kernel)
  B  per channel along axis 0, 4096 EQUAL scales                             (shortrows_kernel: the outputs of C, bit for bit)
  C  per tensor, that scale                                                  (flat_kernel)
  D  per channel along axis 0, two alternating scales                        (shortrows_kernel)
  E  per tensor through the per-channel entry point: ONE channel of n elements (whatever the dispatcher picks)
B against C: the same output bits, only the kernel differs.  A against B: the same kernel, only the parameters / outputs differ.
    python tools/experiments/chanlast2/edge.py [bf16|f16|f32] [rows] [cols]"""
import ctypes
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, REPO)
import torch
from mct_quantizers_amd.hip import native

lib = native.load()
dt_name = sys.argv[1] if len(sys.argv) > 1 else "bf16"
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
cols = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
tdt, dtc, esz = {"f32": (torch.float32, 0, 4), "f16": (torch.float16, 1, 2), "bf16": (torch.bfloat16, 2, 2)}[dt_name]
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
nb = rows * cols * esz * 2
ring = min(64, max(2, -(-(512 << 20) // nb) + 1))
xs = [(torch.randn(rows, cols, device=dev) * 2).to(tdt) for _ in range(ring)]
ys = [torch.empty_like(x) for x in xs]
S = 0.03
s_rand = (torch.rand(rows, device=dev) * 0.05 + 0.01).contiguous()
s_same = torch.full((rows,), S, device=dev)
s_two = torch.where(torch.arange(rows, device=dev) % 2 == 0, torch.tensor(S, device=dev), torch.tensor(0.05, device=dev)).contiguous()
s_one = torch.full((1,), S, device=dev)


def per_channel(s, r=rows, inner=cols):
    def call(i):
        assert lib.mctq_fq_per_channel(xs[i % ring].data_ptr(), ys[i % ring].data_ptr(), 1, r, inner, dtc, s.data_ptr(), None, -128, 127, stream) == 0
    return call


def per_tensor(i):
    assert lib.mctq_fq_per_tensor(xs[i % ring].data_ptr(), ys[i % ring].data_ptr(), rows * cols, dtc, S, 0, -128, 127, stream) == 0


arms = [("A per-channel, 4096 scales", per_channel(s_rand)), ("B per-channel, equal scales", per_channel(s_same)), ("C per-tensor", per_tensor),
        ("D per-channel, two scales", per_channel(s_two)), ("E one channel of n elements", per_channel(s_one, 1, rows * cols))]
names = {}
outs = {}
for name, call in arms:
    call(0)
    torch.cuda.synchronize()
    names[name] = native.last_launch().split("(")[0]
    outs[name] = ys[0].clone()
print("B == C bit for bit:", torch.equal(outs[arms[1][0]].view(torch.int16 if esz == 2 else torch.int32), outs[arms[2][0]].view(torch.int16 if esz == 2 else torch.int32)),
      "| E == C:", torch.equal(outs[arms[4][0]].view(torch.int16 if esz == 2 else torch.int32), outs[arms[2][0]].view(torch.int16 if esz == 2 else torch.int32)))
for name, call in arms:                      # pre-warm every arm
    for i in range(3000):
        call(i)
torch.cuda.synchronize()
times = {name: [] for name, _ in arms}
for rnd in range(5):
    for name, call in arms:
        for i in range(50):
            call(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(200):
            call(i)
        e1.record()
        torch.cuda.synchronize()
        times[name].append(e0.elapsed_time(e1) * 1e3 / 200)
print(f"{dt_name} {rows}x{cols}, {nb >> 20} MiB per launch, ring {ring}")
for name, _ in arms:
    t = sorted(times[name])
    print(f"  {name:32s} {t[0]:6.2f} / {t[2]:6.2f} / {t[-1]:6.2f} us (min / median / max of 5)  frac {nb / t[2] / 8e6:.3f}   [{names[name]}]")
