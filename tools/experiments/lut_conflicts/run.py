#!/usr/bin/env python3
"""Driver of the LDS-layout experiment (lut_conflicts.hip): config 4 (WeightsLUTSymmetric, 16 codes, 4096 x 11008 float32)
through rows_kernel<LutTableXOp<MODE>> for MODE 0..4, bench.py's protocol (1 s pre-warm of the same loop, cold ring, outputs
kept alive, HIP events around 300 launches), modes 0-2 compared bit for bit with the shipped quantizer's output.

    python tools/build_variant.py lut_conflicts
    python tools/experiments/lut_conflicts/run.py                # timing table
    rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE ... -- \\
        python3 tools/experiments/lut_conflicts/run.py pmc       # 40 launches per mode, no timing (kernel names differ by mode)
"""
import ctypes
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, REPO)
import numpy as np
import torch

import mct_quantizers_amd as mq
from mct_quantizers_amd import workloads
from mct_quantizers_amd.hip import native

pmc = len(sys.argv) > 1 and sys.argv[1] == "pmc"
xlib = ctypes.CDLL(os.path.join(REPO, "tools", "ablate", "libmctq_hip_lut_conflicts.so"))
fn = xlib.mctq_x_lutt_per_channel_f32
P, I64, I32, F = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_float
fn.argtypes = [I32, P, P, I64, I64, I64, P, F, P, I32, F, F, F, P]
dev = torch.device("cuda", 0)
x_np = workloads.make_input("cfg4")
wl = workloads.make_workload("cfg4", x_np)
q = getattr(mq.pytorch_quantizers, wl.quantizer)(**wl.kwargs)
x0 = torch.from_numpy(x_np).to(dev)
want = q(x0)
thr = torch.tensor(wl.kwargs["threshold"], dtype=torch.float32, device=dev)
tab_np = native.build_lut_table(workloads.CFG4_LUT, 128.0, -128.0, 127.0)
tab = torch.from_numpy(tab_np).to(dev)
rows, cols = x_np.shape
stream = torch.cuda.current_stream().cuda_stream
ring = 3
xs = [x0] + [x0.clone() for _ in range(ring - 1)]
ys = [torch.empty_like(x0) for _ in range(ring)]
nbytes = x_np.size * 8


def call(mode, i):
    rc = fn(mode, xs[i % ring].data_ptr(), ys[i % ring].data_ptr(), 1, rows, cols, thr.data_ptr(), 1e-8, tab.data_ptr(),
            tab.shape[0] - 1, 128.0, -128.0, 127.0, stream)
    assert rc == 0, rc


names = ["0 shipped layout (ds_read_b64)", "1 swizzled cell index", "2 two 4-byte planes", "3 one cell: NO conflicts (wrong results)",
         "4 T plane only: 4-byte gather (wrong results)"]
for mode in range(5):
    for i in range(ring):
        call(mode, i)
    torch.cuda.synchronize()
    if mode <= 2:
        assert all(torch.equal(y, want) for y in ys), f"mode {mode} differs from the shipped quantizer"
    if pmc:
        for i in range(40):
            call(mode, i)
        torch.cuda.synchronize()
        continue
    t0, n = time.perf_counter(), 0
    while time.perf_counter() - t0 < 1.0:
        call(mode, n); n += 1
        if n % 256 == 0:
            torch.cuda.synchronize()
    res = []
    for rep in range(3):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(300):
            call(mode, i)
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) * 1e3 / 300)
    res.sort()
    print(f"mode {names[mode]:48s} {res[0]:7.2f} {res[1]:7.2f} {res[2]:7.2f} us   {nbytes / res[1] / 1e3:6.0f} GB/s  exact: {mode <= 2}", flush=True)
if not pmc:      # the shipped kernel through the public class, same protocol
    for rep in range(2):
        t0, n = time.perf_counter(), 0
        while time.perf_counter() - t0 < 0.5:
            ys[n % ring] = q(xs[n % ring]); n += 1
            if n % 256 == 0:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(300):
            ys[i % ring] = q(xs[i % ring])
        e1.record(); torch.cuda.synchronize()
        print(f"shipped rows_kernel<LutTableOp> through the class ({native.last_launch()}): {e0.elapsed_time(e1) * 1e3 / 300:7.2f} us", flush=True)
