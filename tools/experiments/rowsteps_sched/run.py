#!/usr/bin/env python3
"""Driver of the rowsteps schedule experiment (rowsteps_sched.hip): per shape and storage type the shipped library's two
choices (rows_kernel, rowsteps_kernel through the tuning key) and the experiment's modes 0-6, bench.py's protocol (0.5 s
pre-warm of the same loop per setting, cold ring, outputs kept alive, HIP events around 200 launches, best and median of 3);
every mode compared bit for bit with the shipped result."""
import ctypes
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, REPO)
import torch

from mct_quantizers_amd.hip import native, ops

xlib = ctypes.CDLL(os.path.join(REPO, "tools", "ablate", "libmctq_hip_rowsteps_sched.so"))
fn = xlib.mctq_x_rowsteps
P, I64, I32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32
fn.argtypes = [I32, P, P, I64, I64, I32, P, I32, I32, I32, P]
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
SHAPES = [("bf16", 4096, 4096), ("bf16", 8192, 2048), ("f32", 4096, 2048), ("f32", 2048, 3072),      # one round of blocks
          ("bf16", 2048, 4096), ("f32", 2048, 2048),                                                     # half a round
          ("bf16", 16384, 4096), ("f32", 8192, 2048), ("bf16", 8192, 6144), ("f32", 16384, 3072)]      # two and more rounds


def timed(call, pre=0.5, n=200):
    t0, k = time.perf_counter(), 0
    while time.perf_counter() - t0 < pre:
        call(k); k += 1
        if k % 256 == 0:
            torch.cuda.synchronize()
    out = []
    for _ in range(3):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(n):
            call(i)
        e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) * 1e3 / n)
    out.sort()
    return out[0], out[1]


print(f"{'shape':22s} {'rows_kernel':>12s} {'rowsteps(lib)':>14s} " + " ".join(f"{'mode ' + str(m):>12s}" for m in range(7)))
for dt_name, rows, cols in SHAPES:
    tdt = torch.bfloat16 if dt_name == "bf16" else torch.float32
    nb = rows * cols * (2 if dt_name == "bf16" else 4) * 2
    ring = max(2, -(-(512 << 20) // nb) + 1)
    xs = [(torch.randn(rows, cols, device=dev) * 2).to(tdt) for _ in range(ring)]
    ys = [torch.empty_like(x) for x in xs]
    s = (torch.rand(rows, device=dev) * 0.05 + 0.01).contiguous()
    z = torch.zeros(rows, dtype=torch.int32, device=dev)
    nt = 2 if nb // 2 <= (32 << 20) else 1
    dtc = 0 if dt_name == "f32" else 2

    def lib_call(i):
        ys[i % ring] = ops.fq_per_channel(xs[i % ring], s, z, 0, -128, 127, True)
    res = []
    for rs in (0, 1):                             # the shipped library: rows_kernel / rowsteps_kernel forced through the tuning key
        native.set_tuning("rowsteps", rs)
        res.append(timed(lib_call))
        want = ops.fq_per_channel(xs[0], s, z, 0, -128, 127, True).clone()
    native.set_tuning("rowsteps", 2)
    cells = [f"{res[0][0]:5.2f}/{res[0][1]:5.2f}", f"{res[1][0]:5.2f}/{res[1][1]:5.2f}"]
    for mode in range(7):
        def call(i, mode=mode):
            rc = fn(mode, xs[i % ring].data_ptr(), ys[i % ring].data_ptr(), rows, cols, dtc, s.data_ptr(), -128, 127, nt, stream)
            assert rc == 0, rc
        ys = [torch.empty_like(x) for x in xs]
        call(0)
        torch.cuda.synchronize()
        assert torch.equal(ys[0], want), f"mode {mode} differs on {dt_name} {rows}x{cols}"
        lo, med = timed(call)
        cells.append(f"{lo:5.2f}/{med:5.2f}")
    print(f"{dt_name + ' ' + str(rows) + 'x' + str(cols):22s} {cells[0]:>12s} {cells[1]:>14s} " + " ".join(f"{c:>12s}" for c in cells[2:]), flush=True)
