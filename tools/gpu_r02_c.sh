#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
timeout 300 python tools/host_overhead.py 2>&1 | grep -v Adjusting > gpurun_out/host_overhead.log
timeout 600 python tools/requant_model_weights.py > gpurun_out/requant_model_weights.log 2>&1
for n in 1 8; do timeout 300 python bench.py --no-cpu --config cfg3 --batch $n --steps 2000 --warmup 100 2>&1 | grep -v Adjusting > gpurun_out/bench_cfg3_n$n.log; done
timeout 600 python tools/compare_aten_gpu.py 2>&1 | grep -v Adjusting > gpurun_out/compare_aten.log
tail -6 gpurun_out/pytest_gpu.log; cat gpurun_out/host_overhead.log gpurun_out/requant_model_weights.log; cut -c1-250 gpurun_out/compare_aten.log
for n in 1 8; do tail -1 gpurun_out/bench_cfg3_n$n.log | cut -c1-400; done
