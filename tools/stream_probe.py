"""Config-3 activation stream at small N: eager holder calls vs CapturedStream (one hipGraph of `depth` kernel nodes over
`lanes` parallel branches).  us per batch by wall clock (host + GPU, synchronised at both ends), outputs compared."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mct_quantizers_amd as mq
from mct_quantizers_amd import workloads
from mct_quantizers_amd.pytorch.graphs import capture_stream

Q = mq.pytorch_quantizers
for n in (1, 8, 64):
    x_np = workloads.make_input("cfg3", batch=n)
    wl = workloads.make_workload("cfg3", x_np)
    holder = mq.PytorchActivationQuantizationHolder(getattr(Q, wl.quantizer)(**wl.kwargs)).cuda()
    x = torch.from_numpy(x_np).cuda()
    want = holder(x)
    reps = 2000 if n <= 8 else 400
    for _ in range(50): holder(x)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): y = holder(x)
    torch.cuda.synchronize(); eager = (time.perf_counter() - t) / reps * 1e6
    print(f"N={n:3d} ({x.numel() * 8 / 1e6:.1f} MB/batch): eager holder call {eager:6.2f} us/batch", flush=True)
    for depth in (1, 4, 16, 20, 64):
        for lanes in (0, 1, 2, 4, 8):                 # 0: the fused batched launch (no graph)
            if lanes > depth: continue
            st = capture_stream(holder, x, depth=depth, lanes=max(1, lanes), mode="fused" if lanes == 0 else "graph")
            for _ in range(5): st.run()
            torch.cuda.synchronize(); t = time.perf_counter()
            r = max(4, reps // depth)
            for _ in range(r): st.run()
            torch.cuda.synchronize(); us = (time.perf_counter() - t) / (r * depth) * 1e6
            same = all(torch.equal(o, want) for o in st.outputs)
            print(f"        depth {depth:3d} {'fused  ' if lanes == 0 else 'lanes ' + str(lanes)}: {us:6.2f} us/batch  {x.numel() * 8 / us / 1e3:7.0f} GB/s  equal={same}", flush=True)
            st.release()
