#!/usr/bin/env python3
"""A / B of two ARMS on the single-tensor C-ABI entry points, both in ONE process on the same buffers, timed alternately
(A B A B): an arm is a library build (the shipped one or a tools/build_variant.py build) plus tuning keys.  bench.py's
protocol per arm (0.4 s pre-warm, cold ring > 512 MiB, outputs kept, HIP events around 100 launches, best / median of 5); outputs
of the two arms compared bit for bit.

    python tools/ab_probe.py --a shipped --b sched1 --cases affine          (B = tools/ablate/libmctq_hip_sched1.so)
    python tools/ab_probe.py --a shipped --b shipped:shortrows=3 --cases affine16,affine32
    python tools/ab_probe.py --a shipped --b lut16wide --cases lut16"""
import argparse, ctypes, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--a", default="shipped")
ap.add_argument("--b", required=True)
ap.add_argument("--cases", default="affine16")
args = ap.parse_args()
P, I64, I32, F32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_float


def arm(spec):
    name, _, tune = spec.partition(":")
    path = os.path.join(REPO, "mct_quantizers_amd", "lib", "libmctq_hip.so") if name == "shipped" else \
        os.path.join(REPO, "tools", "ablate", f"libmctq_hip_{name}.so")
    lib = ctypes.CDLL(path)
    lib.mctq_fq_per_channel.argtypes = [P, P, I64, I64, I64, I32, P, P, I32, I32, P]
    lib.mctq_fq_per_tensor.argtypes = [P, P, I64, I32, F32, I32, I32, I32, P]
    lib.mctq_fq_per_tensor_tqp.argtypes = [P, P, I64, I32, P, P, I32, I32, P]
    lib.mctq_lutt_per_channel.argtypes = [P, P, I64, I64, I64, I32, P, F32, P, I32, F32, F32, F32, P]
    lib.mctq_set_tuning.argtypes = [ctypes.c_char_p, I32]
    lib.mctq_last_launch.restype = ctypes.c_char_p
    keys = [kv.split("=") for kv in tune.split(",") if kv]

    def apply(reset=False):
        for k, v in keys:
            assert lib.mctq_set_tuning(k.encode(), {"shortrows": 1, "paced": 1, "rowsteps": 2, "unroll": 4, "nt": 1}.get(k, 0) if reset else int(v)) == 0, (k, v)
    return spec, lib, apply


A, B = arm(args.a), arm(args.b)
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
DT = {"f32": (torch.float32, 0), "f16": (torch.float16, 1), "bf16": (torch.bfloat16, 2)}


def timed(call, pre=0.4, n=100, reps=5):
    t0, k = time.perf_counter(), 0
    while time.perf_counter() - t0 < pre:
        call(k); k += 1
        if k % 128 == 0:
            torch.cuda.synchronize()
    out = []
    for _ in range(reps):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(n):
            call(i)
        e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) * 1e3 / n)
    out.sort()
    return out[0], out[len(out) // 2]


def run(label, nb, make_call, ys_of):
    """make_call(lib) -> call(i); ys_of() -> list of output tensors of ring slot 0"""
    res, variants, outs = [], [], []
    for rep in range(2):
        for spec, lib, apply in (A, B):
            apply()
            call = make_call(lib)
            if rep == 0:
                for y in ys_of(): y.zero_()
                call(0); torch.cuda.synchronize()
                variants.append(lib.mctq_last_launch().decode().split("(")[0])
                outs.append([y.clone() for y in ys_of()])
            res.append(timed(call))
            apply(reset=True)
    iv = lambda t: t.view(torch.int16 if t.element_size() == 2 else torch.int32)
    same = all(torch.equal(iv(a), iv(b)) for a, b in zip(*outs))
    a_med = min(res[0][1], res[2][1]); b_med = min(res[1][1], res[3][1])
    cells = "  ".join(f"{'AB'[i % 2]} {lo:6.2f}/{med:6.2f}" for i, (lo, med) in enumerate(res))
    print(f"{label:30s} {cells}   B/A {b_med / a_med:.3f}  A {nb / a_med / 8e6:.3f} B {nb / b_med / 8e6:.3f}  equal={same}  "
          f"[A {variants[0]} | B {variants[1]}]", flush=True)


TQP = [("tqp", 4096, 4096), ("pt", 4096, 4096), ("tqp", 2048, 4096), ("pt", 2048, 4096), ("tqp", 1024, 4096), ("tqp", 64, 150528), ("pt", 64, 150528),
       ("tqp", 8, 150528), ("pt", 8, 150528)]
SMALL = [("pc0", 128, 4096), ("pc0", 256, 4096), ("pc0", 512, 4096), ("pc0", 1024, 4096), ("pc0", 1536, 4096), ("pc0", 2048, 4096), ("pc0", 3072, 4096),
         ("pc0", 512, 4608), ("pc0", 64, 65536), ("pc0", 16, 262144), ("pc0", 4, 1048576), ("pc0", 3, 1605632), ("pc0", 8, 2097152), ("pc0", 32, 524288),
         ("pc0", 4096, 256), ("pc0", 16384, 64), ("pc0", 1024, 1024), ("pc0", 256, 1020), ("pc0", 65536, 16), ("pc0", 8192, 1020), ("pc0", 300, 576)]
RAGGED = [("pc0", 16384, 1020), ("pc0", 4096, 4100), ("pc0", 4096, 4099), ("pc0", 8192, 1020), ("pc0", 65536, 252), ("pc0", 32768, 516), ("pc0", 2048, 8190),
          ("pc0", 131072, 100), ("pc0", 1048576, 12), ("pc0", 50257, 772), ("pc0", 300, 1020), ("pc0", 16384, 1024)]
PACED16 = [("pt", r, 4096) for r in (2048, 2560, 2816, 3072, 3328, 3584, 3840, 4096, 4104, 4352, 4608, 8192)]     # 1/2 ... 2 rounds of 16-bit blocks
PACED32 = [("pt", r, 4096) for r in (1024, 1280, 1408, 1536, 1664, 1792, 1920, 2048, 2052, 2304, 4096)]
PACEDROWS32 = [("pc0", 2048, 4096), ("pc0", 1792, 4096), ("pc0", 1536, 4096), ("pc0", 4096, 2048), ("pc0", 2048, 3072), ("pc0", 8192, 1024), ("pc0", 32768, 256),
               ("pc0", 131072, 64), ("pc0", 7168, 1024), ("pc0", 1024, 8192), ("pc0", 8192, 1020), ("pc0", 2056, 4096), ("pc0", 1408, 4096)]
PACEDROWS16 = [("pc0", 4096, 4096), ("pc0", 16384, 1024), ("pc0", 65536, 256), ("pc0", 262144, 64), ("pc0", 14336, 1024), ("pc0", 12288, 1024),
               ("pc0", 1048576, 16), ("pc0", 16384, 1020), ("pc0", 3584, 4096), ("pc0", 4104, 4096)]
ROUNDS = [("pc0", 8192, 2048), ("pc0", 6144, 4096), ("pc0", 8192, 4096), ("pc0", 12288, 4096), ("pc0", 16384, 4096), ("pc0", 4096, 2048),
          ("pc0", 2048, 3072), ("pc0", 1024, 4096), ("pc0", 512, 4096)]


def affine_cases(dts, shapes=None):
    shapes = shapes or [("pc0", 4096, 4096), ("pc0", 16384, 1024), ("pc0", 16384, 1020), ("pc0", 65536, 256), ("pc0", 1048576, 16),
              ("pc0", 4096, 4100), ("pc0", 256, 65536), ("pc0", 8192, 8192), ("pc0", 50257, 768), ("pc0", 2048, 4608),
              ("pc1", 4096, 4096), ("pc1", 200704, 256), ("pt", 4096, 4096), ("pt", 9633792, 8), ("pt", 1024, 4096)]
    for dt in dts:
        tdt, dtc = DT[dt]
        for kind, rows, cols in shapes:
            esz = 2 if dt != "f32" else 4
            nb = rows * cols * esz * 2
            ring = min(64, max(2, -(-(512 << 20) // nb) + 1))
            xs = [(torch.randn(rows, cols, device=dev) * 2).to(tdt) for _ in range(ring)]
            ys = [torch.empty_like(x) for x in xs]
            C = rows if kind == "pc0" else cols
            s = (torch.rand(C, device=dev) * 0.05 + 0.01)
            s1 = torch.tensor([0.031], device=dev)
            z1 = torch.tensor([3], dtype=torch.int32, device=dev)

            def make_call(lib):
                if kind == "pc0":
                    return lambda i: lib.mctq_fq_per_channel(xs[i % ring].data_ptr(), ys[i % ring].data_ptr(), 1, rows, cols, dtc, s.data_ptr(), None, -128, 127, stream)
                if kind == "pc1":
                    return lambda i: lib.mctq_fq_per_channel(xs[i % ring].data_ptr(), ys[i % ring].data_ptr(), rows, cols, 1, dtc, s.data_ptr(), None, -128, 127, stream)
                if kind == "tqp":
                    return lambda i: lib.mctq_fq_per_tensor_tqp(xs[i % ring].data_ptr(), ys[i % ring].data_ptr(), rows * cols, dtc, s1.data_ptr(), z1.data_ptr(), 0, 255, stream)
                return lambda i: lib.mctq_fq_per_tensor(xs[i % ring].data_ptr(), ys[i % ring].data_ptr(), rows * cols, dtc, 0.031, 3, 0, 255, stream)
            run(f"{dt} {kind} {rows}x{cols}", nb, make_call, lambda: [ys[0]])
            del xs, ys
            torch.cuda.empty_cache()


def lut16_cases():
    from mct_quantizers_amd.hip import native
    lutv = np.asarray([-128, -96, -64, -40, -24, -12, -5, 0, 5, 12, 24, 40, 64, 96, 120, 127], dtype=np.float32)
    table = torch.from_numpy(native.build_lut_table(lutv, 128.0, -128.0, 127.0)).to(dev)
    entries = table.shape[0] - 1
    for dt in ("bf16", "f16", "f32"):
        tdt, dtc = DT[dt]
        for rows, cols in ((4096, 11008), (4096, 4096), (16384, 1024)):
            esz = 2 if dt != "f32" else 4
            nb = rows * cols * (esz + 4)
            ring = min(16, max(2, -(-(512 << 20) // nb) + 1))
            xs = [torch.randn(rows, cols, device=dev).to(tdt) for _ in range(ring)]
            ys = [torch.empty(rows, cols, device=dev, dtype=torch.float32) for _ in range(ring)]
            thr = (torch.rand(rows, device=dev) + 3.5)

            def make_call(lib):
                return lambda i: lib.mctq_lutt_per_channel(xs[i % ring].data_ptr(), ys[i % ring].data_ptr(), 1, rows, cols, dtc, thr.data_ptr(), 1e-8,
                                                           table.data_ptr(), entries, 128.0, -128.0, 127.0, stream)
            run(f"lutt {dt} {rows}x{cols}", nb, make_call, lambda: [ys[0]])
            del xs, ys
            torch.cuda.empty_cache()


print(f"A = {A[0]}   B = {B[0]}   (us best/median per arm, A B A B; B/A on the better median of each arm)", flush=True)
for c in args.cases.split(","):
    if c == "affine16": affine_cases(["bf16", "f16"])
    elif c == "affinebf16": affine_cases(["bf16"])
    elif c == "affine32": affine_cases(["f32"])
    elif c == "tqp16": affine_cases(["bf16", "f16"], TQP)
    elif c == "tqp32": affine_cases(["f32"], TQP)
    elif c == "pacedrows32": affine_cases(["f32"], PACEDROWS32)
    elif c == "pacedrows16": affine_cases(["bf16", "f16"], PACEDROWS16)
    elif c == "paced16": affine_cases(["bf16", "f16"], PACED16)
    elif c == "paced32": affine_cases(["f32"], PACED32)
    elif c == "small16": affine_cases(["bf16"], SMALL)
    elif c == "small32": affine_cases(["f32"], SMALL)
    elif c == "ragged16": affine_cases(["bf16", "f16"], RAGGED)
    elif c == "ragged32": affine_cases(["f32"], [("pc0", 1048576, 13), ("pc0", 2097152, 7), ("pc0", 1048576, 16)])
    elif c == "rounds16": affine_cases(["bf16"], ROUNDS)
    elif c == "rounds32": affine_cases(["f32"], ROUNDS)
    elif c == "lut16": lut16_cases()
    else: raise SystemExit(f"unknown case set {c}")
