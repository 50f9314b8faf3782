"""Where do the ~40 us go that the host wall clock of a 20-step timed region carries beyond 20 x the kernel period?
Stamps inside bench.py's region protocol (sync -> t0 -> [ev0, K launches, ev_first after launch 1, ev1] -> spin on ev1 ->
torch.cuda.synchronize -> t1) for config 2, K = 20 and 200."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mct_quantizers_amd as mq
from mct_quantizers_amd import workloads

x_np = workloads.make_input("cfg2")
wl = workloads.make_workload("cfg2", x_np)
q = getattr(mq.pytorch_quantizers, wl.quantizer)(**wl.kwargs)
xs = [torch.from_numpy(x_np).cuda() for _ in range(5)]
ys = [None] * 5
for i in range(2000):
    ys[i % 5] = q(xs[i % 5])
torch.cuda.synchronize()
for K in (20, 200):
    for variant in ("events ev0+ev_first+ev1 (bench.py)", "only ev1", "no events, synchronize only"):
        rows = []
        for rep in range(30):
            torch.cuda.synchronize()
            e0, ef, e1 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            t0 = time.perf_counter()
            if variant.startswith("events"):
                e0.record()
            for i in range(K):
                ys[i % 5] = q(xs[i % 5])
                if i == 0 and variant.startswith("events"):
                    ef.record()
            t_issued = time.perf_counter()
            if not variant.startswith("no events"):
                e1.record()
                while not e1.query():
                    pass
            t_spin = time.perf_counter()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            dev = (ef.elapsed_time(e1) * 1e3 / (K - 1)) if variant.startswith("events") else float("nan")
            rows.append(((t1 - t0) * 1e6, (t_issued - t0) * 1e6, (t_spin - t0) * 1e6, (t1 - t_spin) * 1e6, dev))
        rows.sort()
        w, iss, spin, sync, dev = rows[len(rows) // 2]
        print(f"K={K:4d} {variant:38s} wall {w:8.1f} us = {w / K:6.2f}/step | launches issued at {iss:7.1f} | completion seen at {spin:8.1f} | "
              f"synchronize +{sync:5.1f} | steady kernel period {dev:6.2f}")
