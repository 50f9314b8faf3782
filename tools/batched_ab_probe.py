#!/usr/bin/env python3
"""A / B of two builds of libmctq_hip.so on the table-driven batched launch (mctq_fq_batch_pack / mctq_fq_batch_run), pure ctypes,
both libraries loaded into ONE process and timed alternately (A B A B ...) on the same buffers: all weight tensors of ResNet-50
(54 tensors, 204 MB of traffic per launch in float32), the same list in bfloat16, 16 x 4096^2 (long rows), and the gather launch of
a few short-row float32 shapes.  bench.py's protocol per arm: pre-warm 0.4 s, cold ring (> 512 MiB between two uses of a buffer),
outputs kept, HIP events around 100 launches, best / median of 5.

    python tools/build_variant.py batch_exact -DMCTQ_BATCH_EXACT_RECIP=2 --units=mctq_batched.hip   (B arm: recip_exact in the list launches too)
    python tools/batched_ab_probe.py [A.so] [B.so]        (defaults: the shipped library, tools/ablate/libmctq_hip_batch_exact.so)"""
import ctypes, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
import torch
from mct_quantizers_amd import workloads
from mct_quantizers_amd.hip import native

paths = sys.argv[1:3] if len(sys.argv) >= 3 else [os.path.join(REPO, "mct_quantizers_amd", "lib", "libmctq_hip.so"),
                                                    os.path.join(REPO, "tools", "ablate", "libmctq_hip_batch_exact.so")]
libs = []
P, I64, I32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32
for p in paths:
    lib = ctypes.CDLL(p)
    lib.mctq_fq_batch_pack.restype = I64
    lib.mctq_fq_batch_pack.argtypes = [P, I32, P, I64]
    lib.mctq_fq_batch_run.argtypes = [P, P, P]
    lib.mctq_fq_per_channel.argtypes = [P, P, I64, I64, I64, I32, P, P, I32, I32, P]
    lib.mctq_last_launch.restype = ctypes.c_char_p
    lib.mctq_build_id.restype = ctypes.c_char_p
    libs.append(lib)
print("A =", paths[0], libs[0].mctq_build_id().decode(), "\nB =", paths[1], libs[1].mctq_build_id().decode(), flush=True)
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream


def timed(call, pre=0.4, n=100, reps=5):
    t0, k = time.perf_counter(), 0
    while time.perf_counter() - t0 < pre:
        call(k); k += 1
        if k % 64 == 0:
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


def model_list(tdt, dtc, name, tensors):
    """tensors: list of (numpy weight, per-channel thresholds) quantized along axis 0"""
    nbytes = sum(int(w.size) for w, _ in tensors) * torch.empty(0, dtype=tdt).element_size() * 2
    ring = min(16, max(2, -(-(512 << 20) // nbytes) + 1))
    sets = []
    scales = [torch.tensor(np.asarray(t, dtype=np.float64) / 128.0, dtype=torch.float32, device=dev) for _, t in tensors]
    for r in range(ring):
        xs = [torch.from_numpy(w).to(dev).to(tdt) for w, _ in tensors]
        ys = [torch.empty_like(x) for x in xs]
        items = (native.FqItem * len(xs))()
        for i, (x, y, s) in enumerate(zip(xs, ys, scales)):
            c = x.shape[0]
            items[i] = native.FqItem(x.data_ptr(), y.data_ptr(), 1, c, x.numel() // c, s.data_ptr(), None, -128, 127, dtc, 0)
        tabs = []
        for lib in libs:
            need = lib.mctq_fq_batch_pack(items, len(xs), None, 0)
            host = (ctypes.c_uint8 * need)()
            assert lib.mctq_fq_batch_pack(items, len(xs), host, need) == need
            devt = torch.frombuffer(bytearray(host), dtype=torch.uint8).to(dev)
            tabs.append((host, devt))
        sets.append((xs, ys, items, tabs))
    res = []
    for rep in range(2):                                  # A B A B
        for k, lib in enumerate(libs):
            def call(i, lib=lib, k=k):
                _, _, _, tabs = sets[i % ring]
                assert lib.mctq_fq_batch_run(tabs[k][0], tabs[k][1].data_ptr(), stream) == 0
            res.append((k, timed(call)))
    # bit equality of the two builds' outputs on set 0
    outs = []
    for k, lib in enumerate(libs):
        xs, ys, _, tabs = sets[0]
        for y in ys: y.zero_()
        assert lib.mctq_fq_batch_run(tabs[k][0], tabs[k][1].data_ptr(), stream) == 0
        torch.cuda.synchronize()
        outs.append([y.clone() for y in ys])
    same = all(torch.equal(a.view(torch.int16 if a.element_size() == 2 else torch.int32), b.view(torch.int16 if b.element_size() == 2 else torch.int32))
               for a, b in zip(*outs))
    cells = "  ".join(f"{'AB'[k]} {lo:6.2f}/{med:6.2f} ({nbytes / med / 8e6:.3f})" for k, (lo, med) in res)
    print(f"{name:34s} {nbytes / 1e6:7.1f} MB  {cells}  equal={same}  [{libs[0].mctq_last_launch().decode()}]", flush=True)
    del sets
    torch.cuda.empty_cache()


weights = workloads.make_model_weights("resnet50")
res50 = [(w, kw["threshold"]) for w, kw in weights]
model_list(torch.float32, 0, "resnet50 54 weights f32", res50)
model_list(torch.bfloat16, 2, "resnet50 54 weights bf16", res50)
rng = np.random.default_rng(0)
lin = [(rng.standard_normal((4096, 4096), dtype=np.float32), [1.0 + 0.01 * (i % 97) for i in range(4096)]) for _ in range(4)]
model_list(torch.float32, 0, "4 x 4096^2 f32 (rows in SGPRs)", lin)
for shape in ((512, 512, 3, 3), (256, 256, 3, 3), (1024, 256, 1, 1), (2048, 512, 1, 1), (64, 64, 1, 1)):
    c = shape[0]
    k = max(1, (32 << 20) // (int(np.prod(shape)) * 4))
    many = [(rng.standard_normal(shape, dtype=np.float32), [1.0 + 0.01 * (i % 97) for i in range(c)]) for _ in range(min(k, 64))]
    model_list(torch.float32, 0, f"{len(many)} x {shape} f32", many)

print("\nsingle tensors, float32 short rows (gather launch):", flush=True)
for rows, inner in ((16384, 1020), (65536, 256), (262144, 64), (50257, 768), (4096, 4100)):
    nb = rows * inner * 8
    ring = max(2, -(-(512 << 20) // nb) + 1)
    xs = [torch.randn(rows, inner, device=dev) for _ in range(ring)]
    ys = [torch.empty_like(x) for x in xs]
    s = (torch.rand(rows, device=dev) * 0.05 + 0.01)
    res = []
    for rep in range(2):
        for k, lib in enumerate(libs):
            def call(i, lib=lib):
                assert lib.mctq_fq_per_channel(xs[i % ring].data_ptr(), ys[i % ring].data_ptr(), 1, rows, inner, 0, s.data_ptr(), None, -128, 127, stream) == 0
            res.append((k, timed(call)))
    cells = "  ".join(f"{'AB'[k]} {lo:6.2f}/{med:6.2f} ({nb / med / 8e6:.3f})" for k, (lo, med) in res)
    print(f"f32 {rows}x{inner:5d} {cells}  [{libs[0].mctq_last_launch().decode()}]", flush=True)
    del xs, ys
    torch.cuda.empty_cache()
