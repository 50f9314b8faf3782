#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
cd /tmp && rm -rf /tmp/prof1 && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof1 -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --steps 400 > $GRAFT_REPO_ROOT/gpurun_out/rocprof_bench.log 2>&1
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof1 && find /tmp/prof1 -name "*stats*.csv" -exec cp {} gpurun_out/prof1/ \;
find /tmp/prof1 -name "*kernel_trace.csv" | head -1 | xargs -I{} sh -c 'head -1 {} > gpurun_out/prof1/kernel_trace_head.csv; grep rows_kernel {} | tail -50 >> gpurun_out/prof1/kernel_trace_head.csv'
python - <<'PY' > gpurun_out/cpu_threads.log 2>&1
import time, torch, numpy as np, sys, os
sys.path.insert(0, os.getcwd())
from mct_quantizers_amd import workloads
from oracle import torch_cpu
x_np = workloads.make_input("cfg2"); wl = workloads.make_workload("cfg2", x_np)
f = torch_cpu.prepare(wl.quantizer, wl.kwargs); x = torch.from_numpy(x_np)
print("cpu_count", os.cpu_count())
for th in (1, 8, 16, 32, 64, 128, 256):
    torch.set_num_threads(th); f(x)
    t=time.perf_counter(); n=0
    while time.perf_counter()-t < 2.0: f(x); n+=1
    print(th, "threads ms/call", (time.perf_counter()-t)*1e3/n)
PY
tail -5 gpurun_out/pytest_gpu.log; cat gpurun_out/prof1/*stats*.csv | head -20; cat gpurun_out/cpu_threads.log; tail -2 gpurun_out/rocprof_bench.log | cut -c1-600
