#!/bin/bash
# round 2, first GPU pass: suite + where-does-the-fixed-cost-go experiment + host overhead + ATen comparison
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout 300 ./tools/kbench_timeline.bin 200 > gpurun_out/kbench_timeline.log 2>&1; echo "timeline rc=$?" >> gpurun_out/kbench_timeline.log
timeout 300 python tools/host_overhead.py > gpurun_out/host_overhead.log 2>&1; echo "rc=$?" >> gpurun_out/host_overhead.log
MCTQ_BINDING=ctypes timeout 300 python tools/host_overhead.py > gpurun_out/host_overhead_ctypes.log 2>&1
timeout 600 python tools/compare_aten_gpu.py > gpurun_out/compare_aten.log 2>&1; echo "rc=$?" >> gpurun_out/compare_aten.log
timeout 600 python bench.py --no-cpu > gpurun_out/bench.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench.log
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -5 gpurun_out/pytest_gpu.log; tail -40 gpurun_out/kbench_timeline.log; cat gpurun_out/host_overhead.log; cat gpurun_out/compare_aten.log | cut -c1-230; tail -1 gpurun_out/bench.log | cut -c1-600
