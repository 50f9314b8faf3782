"""The config-4 tensor (4096 x 11008 float32) through the LUT decision-table kernel and, same tensor and launch grid
family, through the affine kernel: N launches each, cold ring of inputs AND outputs (the outputs of the last `ring` launches
stay alive, as in bench.py: a dropped output would hand the next launch the same buffer, whose lines then sit in the 256 MiB
Infinity Cache -- MCTQ_PROBE_WARM_OUT=1 gives that half-warm protocol, which round 4's first ablations were taken under).  Run plain for timings, or under
`rocprofv3 --kernel-trace --pmc ...` (tools/gpu_r04_lut_pmc.sh) for per-kernel counters.
    python tools/lut_vs_affine.py [launches] [heavy_unroll]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mct_quantizers_amd as mq
from mct_quantizers_amd import workloads
from mct_quantizers_amd.hip import native
Q = mq.pytorch_quantizers
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
hu = int(sys.argv[2]) if len(sys.argv) > 2 else 0
x_np = workloads.make_input("cfg4"); wl = workloads.make_workload("cfg4", x_np)
ring = 3
xs = [torch.from_numpy(x_np).cuda() for _ in range(ring)]
qlut = getattr(Q, wl.quantizer)(**wl.kwargs)
qa = Q.WeightsSymmetricInferableQuantizer(8, wl.kwargs["threshold"], True, 0)
if hu:
    native.set_tuning("heavy_unroll", hu)
nbytes = x_np.size * 8
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.5:
    for i in range(10): qlut(xs[i % ring]); qa(xs[i % ring])
    torch.cuda.synchronize()
for name, q in (("lut", qlut), ("affine", qa), ("lut", qlut), ("affine", qa)):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    keep = [None] * ring
    warm_out = os.environ.get("MCTQ_PROBE_WARM_OUT", "0") == "1"
    for i in range(n):
        y = q(xs[i % ring])
        if not warm_out: keep[i % ring] = y
    e1.record(); torch.cuda.synchronize()
    del keep, y
    us = e0.elapsed_time(e1) * 1e3 / n
    print(f"{name:7s} {us:7.2f} us  {nbytes / us / 1e3:6.0f} GB/s  frac {nbytes / us / 1e3 / 8000:.3f}  {native.last_launch()}", flush=True)
