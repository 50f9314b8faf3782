#!/usr/bin/env python3
"""Does a consumer that runs right after the channel-last quantizer gain from cached stores (nt = 2) what the quantizer loses to
them?  Per shape and store policy: the quantizer alone, and the pair quantizer -> torch.mm reading the fresh output (cold inputs:
a ring of input buffers; the output is consumed at once, as a wrapped layer does).  Experiment library (chanlast2), mode 2."""
import ctypes, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, REPO)
import torch
xlib = ctypes.CDLL(os.path.join(REPO, "tools", "ablate", "libmctq_hip_chanlast2.so"))
P, I64, I32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32
xlib.mctq_x_lastaxis.argtypes = [I32, I32, I32, P, P, I64, I64, I32, P, P, I32, I32, I32, P]
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream


def timed(call, pre=0.4, n=200):
    t0, k = time.perf_counter(), 0
    while time.perf_counter() - t0 < pre:
        call(k); k += 1
        if k % 128 == 0:
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
    return sorted(out)[1]


for rows, C, m in ((4096, 4096, 64), (4096, 4096, 1024), (2048, 4096, 64), (1024, 4096, 64), (50176, 256, 64), (12544, 256, 64)):
    nb = rows * C * 2 * 2
    ring = min(64, max(3, -(-(512 << 20) // nb) + 1))
    xs = [(torch.randn(rows, C, device=dev) * 2).to(torch.bfloat16) for _ in range(ring)]
    ys = [torch.empty_like(x) for x in xs[:3]]
    s = (torch.rand(C, device=dev) * 0.05 + 0.01).contiguous()
    a = torch.randn(m, C, device=dev).to(torch.bfloat16)
    outs = [torch.empty(m, rows, device=dev, dtype=torch.bfloat16) for _ in range(3)]
    line = f"bf16 {rows}x{C} ({rows * C * 2 >> 20} MiB out), consumer mm [{m}x{C}] x y^T: "
    for nt in (1, 2):
        def q(i):
            assert xlib.mctq_x_lastaxis(2, 2, 1, xs[i % ring].data_ptr(), ys[i % 3].data_ptr(), rows, C, 2, s.data_ptr(), None, -128, 127, nt, stream) == 0
        def pair(i):
            q(i)
            torch.mm(a, ys[i % 3].t(), out=outs[i % 3])
        def mm_only(i):
            torch.mm(a, ys[i % 3].t(), out=outs[i % 3])
        tq, tp = timed(q), timed(pair)
        line += f" nt={nt}: quantizer {tq:6.2f} us, pair {tp:6.2f} us, pair - quantizer {tp - tq:6.2f} |"
    print(line + f" mm alone (warm y) {timed(mm_only):6.2f} us", flush=True)
    del xs, ys
    torch.cuda.empty_cache()
