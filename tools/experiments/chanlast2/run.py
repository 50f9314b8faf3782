#!/usr/bin/env python3
"""Driver of the chanlast2 experiment (chanlast2.hip): per shape and storage type the shipped launch (mctq_fq_per_channel of the same
library build) and the experiment's modes, bench.py's protocol (0.5 s pre-warm of the same loop per setting, cold ring, outputs kept
alive, HIP events around 200 launches, best and median of 3); every exact mode compared bit for bit with the shipped result.
    python tools/build_variant.py chanlast2 && python tools/experiments/chanlast2/run.py [lastaxis|shortrows|recip] ..."""
import ctypes
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, REPO)
import torch

xlib = ctypes.CDLL(os.path.join(REPO, "tools", "ablate", "libmctq_hip_chanlast2.so"))
P, I64, I32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32
xlib.mctq_fq_per_channel.argtypes = [P, P, I64, I64, I64, I32, P, P, I32, I32, P]
xlib.mctq_x_lastaxis.argtypes = [I32, I32, I32, P, P, I64, I64, I32, P, P, I32, I32, I32, P]
xlib.mctq_x_shortrows.argtypes = [I32, P, P, I64, I64, I64, I32, P, P, I32, I32, I32, P]
xlib.mctq_x_recip_check.argtypes = [P, P]
xlib.mctq_last_launch.restype = ctypes.c_char_p
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
DT = {"f32": (torch.float32, 0, 4), "f16": (torch.float16, 1, 2), "bf16": (torch.bfloat16, 2, 2)}
what = set(sys.argv[1:]) or {"recip", "lastaxis", "shortrows"}
if "sched" in what or "contig" in what or "lds" in what or "bisect" in what or "paced" in what:
    what.add("lastaxis")
if "oneround" in what:
    what.add("shortrows")


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


if "recip" in what:
    out = torch.zeros(3, dtype=torch.int64, device=dev)
    assert xlib.mctq_x_recip_check(out.data_ptr(), stream) == 0
    torch.cuda.synchronize()
    bad, first, seen = [int(v) for v in out.cpu()]
    print(f"recip_nr2 vs IEEE 1.0f/d over all float32 bit patterns with 2^-100 <= |d| <= 2^100: checked {seen}, mismatches {bad}"
          + (f", one at bits 0x{first - 1:08x}" if bad else ""), flush=True)


def setup(dt_name, rows, cols, C, with_zp):
    tdt, dtc, esz = DT[dt_name]
    nb = rows * cols * esz * 2
    ring = min(64, max(2, -(-(512 << 20) // nb) + 1))
    xs = [(torch.randn(rows, cols, device=dev) * 2).to(tdt) for _ in range(ring)]
    s = (torch.rand(C, device=dev) * 0.05 + 0.01).contiguous()
    z = (torch.randint(-3, 4, (C,), dtype=torch.int32, device=dev) if with_zp else None)
    nt = 2 if nb // 2 <= (32 << 20) else 1
    return tdt, dtc, nb, ring, xs, s, z, nt


def run_case(label, nb, ring, xs, lib_call, modes):
    ys = [torch.empty_like(x) for x in xs]
    lib_call(0, ys)
    torch.cuda.synchronize()
    want = ys[0].clone()
    variant = xlib.mctq_last_launch().decode().split("(")[0]
    lo, med = timed(lambda i: lib_call(i, ys))
    cells = [f"lib {lo:5.2f}/{med:5.2f} ({nb / med / 8e6:.3f})"]
    for name, fn, exact in modes:
        ys = [torch.empty_like(x) for x in xs]
        rc = fn(0, ys)
        if rc != 0:
            cells.append(f"{name} n/a")
            continue
        torch.cuda.synchronize()
        same = torch.equal(ys[0].view(torch.int16 if ys[0].element_size() == 2 else torch.int32),
                           want.view(torch.int16 if want.element_size() == 2 else torch.int32))
        if exact:
            assert same, f"{name} differs on {label}"
        lo, med = timed(lambda i: fn(i, ys))
        cells.append(f"{name} {lo:5.2f}/{med:5.2f}" + ("" if exact else "*"))
    print(f"{label:28s} " + " | ".join(cells) + f"   [{variant}]", flush=True)


if "lastaxis" in what:
    print("\n== lastaxis (channels on the fastest axis); us best/median, (frac of 8 TB/s); * = timing-only mode ==", flush=True)
    cases = [("bf16", 4096, 4096, False), ("bf16", 4096, 4096, True), ("f16", 4096, 4096, False), ("f32", 4096, 4096, False),
             ("bf16", 2048, 8192, False), ("bf16", 8192, 8192, False), ("bf16", 200704, 256, False), ("f32", 200704, 256, False),
             ("bf16", 65536, 200, False), ("f32", 65536, 200, False), ("bf16", 1048576, 16, False), ("bf16", 50257, 768, False),
             ("bf16", 50176, 1024, False), ("bf16", 4099, 4096, True)]
    for dt_name, rows, C, with_zp in cases:
        tdt, dtc, nb, ring, xs, s, z, nt = setup(dt_name, rows, C, C, with_zp)
        zp = z.data_ptr() if z is not None else None

        def lib_call(i, ys):
            rc = xlib.mctq_fq_per_channel(xs[i % ring].data_ptr(), ys[i % ring].data_ptr(), rows, C, 1, dtc, s.data_ptr(), zp, -128, 127, stream)
            assert rc == 0

        def mk(mode, u, loops):
            return lambda i, ys: xlib.mctq_x_lastaxis(mode, u, loops, xs[i % ring].data_ptr(), ys[i % ring].data_ptr(), rows, C, dtc,
                                                      s.data_ptr(), zp, -128, 127, nt, stream)
        if "contig" in what:
            modes = [("slab U2", mk(2, 2, 1), True), ("slab U4", mk(2, 4, 1), True), ("contig U2", mk(21, 2, 1), True), ("contig U4", mk(21, 4, 1), True),
                     ("contig repeat", mk(22, 4, 1), True)]
        elif "paced" in what:
            modes = [("U2end", mk(11, 2, 1), True), ("U4end", mk(11, 4, 1), True), ("U2asgo", mk(2, 2, 1), True), ("U4asgo", mk(2, 4, 1), True),
                     ("U2paced", mk(14, 2, 1), True), ("U4paced", mk(14, 4, 1), True), ("U2wait+paced", mk(15, 2, 1), True), ("U4wait+paced", mk(15, 4, 1), True)]
        elif "bisect" in what:
            modes = [("U2", mk(2, 2, 1), True), ("U4", mk(2, 4, 1), True), ("U2 norcp", mk(3, 2, 1), False), ("U2 1dword", mk(6, 2, 1), False),
                     ("U4 1dword", mk(6, 4, 1), False), ("U2 notable", mk(7, 2, 1), False), ("U4 notable", mk(7, 4, 1), False)]
        elif "lds" in what:
            def mkn(mode, u, ntx, lv=1):
                return lambda i, ys: xlib.mctq_x_lastaxis(mode, u, lv, xs[i % ring].data_ptr(), ys[i % ring].data_ptr(), rows, C, dtc,
                                                          s.data_ptr(), zp, -128, 127, ntx, stream)
            modes = [("U2end", mk(11, 2, 1), True), ("U4end", mk(11, 4, 1), True), ("lds U2", mk(31, 2, 1), True), ("lds U4", mk(31, 4, 1), True),
                     ("ldsinv U2", mk(32, 2, 1), True), ("ldsinv U4", mk(32, 4, 1), True), ("lds U2 s2048", mk(31, 2, 2048), True),
                     ("lds U4 s2048", mk(31, 4, 2048), True), (f"lds U2 nt{3 - nt}", mkn(31, 2, 3 - nt), True), (f"lds U4 nt{3 - nt}", mkn(31, 4, 3 - nt), True)]
        elif "sched" in what:
            def mkn(mode, u, ntx):
                return lambda i, ys: xlib.mctq_x_lastaxis(mode, u, 1, xs[i % ring].data_ptr(), ys[i % ring].data_ptr(), rows, C, dtc,
                                                          s.data_ptr(), zp, -128, 127, ntx, stream)
            modes = [("U4", mk(2, 4, 1), True), ("U2", mk(2, 2, 1), True), ("U4end", mk(11, 4, 1), True), ("U2end", mk(11, 2, 1), True),
                     ("U4rot", mk(12, 4, 1), True), ("U2rot", mk(12, 2, 1), True), ("U4gx", mk(13, 4, 1), True), ("U2gx", mk(13, 2, 1), True),
                     (f"U4nt{3 - nt}", mkn(2, 4, 3 - nt), True), (f"U2nt{3 - nt}", mkn(2, 2, 3 - nt), True)]
        else:
            modes = [("early", mk(1, 4, 1), True), ("+rcp", mk(2, 4, 1), True), ("norcp", mk(3, 4, 1), False), ("+rcpU2", mk(2, 2, 1), True),
                     ("loop2", mk(4, 4, 2), True), ("loop4", mk(4, 4, 4), True), ("pipe2", mk(5, 4, 2), True), ("pipe4", mk(5, 4, 4), True),
                     ("pipeU2x4", mk(5, 2, 4), True), ("pipeU2x8", mk(5, 2, 8), True)]
        run_case(f"{dt_name} {rows}x{C}" + (" zp" if with_zp else ""), nb, ring, xs, lib_call, modes)
        del xs
        torch.cuda.empty_cache()

if "sched" in what or "bisect" in what:
    print("\n== reference launches of the same byte count on this box (shipped library) ==", flush=True)
    xlib.mctq_fq_per_tensor.argtypes = [P, P, I64, I32, ctypes.c_float, I32, I32, I32, P]
    for dt_name in ("bf16", "f32"):
        tdt, dtc, nb, ring, xs, s, z, nt = setup(dt_name, 4096, 4096, 4096, False)
        ys = [torch.empty_like(x) for x in xs]
        def rows_call(i):
            assert xlib.mctq_fq_per_channel(xs[i % ring].data_ptr(), ys[i % ring].data_ptr(), 1, 4096, 4096, dtc, s.data_ptr(), None, -128, 127, stream) == 0
        def flat_call(i):
            assert xlib.mctq_fq_per_tensor(xs[i % ring].data_ptr(), ys[i % ring].data_ptr(), 4096 * 4096, dtc, 0.03, 0, -128, 127, stream) == 0
        for name, call in (("per-channel axis 0", rows_call), ("per-tensor", flat_call)):
            call(0); torch.cuda.synchronize()
            variant = xlib.mctq_last_launch().decode().split("(")[0]
            lo, med = timed(call)
            print(f"{dt_name} 4096x4096 {name:20s} {lo:5.2f}/{med:5.2f} ({nb / med / 8e6:.3f})  [{variant}]", flush=True)
        del xs, ys
        torch.cuda.empty_cache()

if "shortrows" in what:
    print("\n== short rows (per-channel along axis 0, inner >= N); modes: data-first IEEE / data-first rcp / params-first IEEE / params-first rcp ==", flush=True)
    cases = [("bf16", 1048576, 16, False), ("bf16", 16384, 1020, False), ("bf16", 16384, 1020, True), ("bf16", 16384, 1024, False),
             ("bf16", 65536, 256, False), ("bf16", 262144, 64, False), ("bf16", 4096, 4100, False), ("bf16", 4096, 4099, False),
             ("f16", 1048576, 16, False), ("f32", 1048576, 16, False), ("f32", 16384, 1020, False), ("f32", 262144, 64, False),
             ("f32", 4096, 4100, False), ("bf16", 50257, 768, False)]
    if "oneround" in what:              # per-channel launches of 3/4 ... 1 round of resident blocks (the window of flat_paced_kernel)
        cases = [("bf16", 4096, 4096, False), ("f16", 4096, 4096, False), ("bf16", 3584, 4096, False), ("bf16", 16384, 1024, False),
                 ("bf16", 65536, 256, False), ("f32", 2048, 4096, False), ("f32", 4096, 2048, False), ("f32", 8192, 1024, False),
                 ("f32", 1792, 4096, False), ("f32", 32768, 256, False)]
    for dt_name, rows, inner, with_zp in cases:
        tdt, dtc, nb, ring, xs, s, z, nt = setup(dt_name, rows, inner, rows, with_zp)
        zp = z.data_ptr() if z is not None else None

        def lib_call(i, ys):
            rc = xlib.mctq_fq_per_channel(xs[i % ring].data_ptr(), ys[i % ring].data_ptr(), 1, rows, inner, dtc, s.data_ptr(), zp, -128, 127, stream)
            assert rc == 0

        def mk(mode):
            return lambda i, ys: xlib.mctq_x_shortrows(mode, xs[i % ring].data_ptr(), ys[i % ring].data_ptr(), rows, inner, rows, dtc,
                                                       s.data_ptr(), zp, -128, 127, nt, stream)
        modes = [("d/ieee", mk(1), True), ("d/rcp", mk(2), True), ("p/ieee", mk(3), True), ("p/rcp", mk(4), True), ("d/rcp paced", mk(5), True), ("d/rcp wait+paced", mk(6), True), ("d/rcp apart+wait+paced", mk(7), True)]
        run_case(f"{dt_name} {rows}x{inner}" + (" zp" if with_zp else ""), nb, ring, xs, lib_call, modes)
        del xs
        torch.cuda.empty_cache()
