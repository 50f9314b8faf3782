#!/bin/bash
# round check: full GPU suite, smoke, default bench (+ rocprof stats of the same command), driver-style short bench, torchrun launch check
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
timeout 900 python bench.py > gpurun_out/bench.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu > gpurun_out/bench_20.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench_20.log
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --gather --steps 100 --warmup 10 --no-cpu > gpurun_out/bench_torchrun.log 2>&1; echo "torchrun rc=$?" >> gpurun_out/bench_torchrun.log
cd /tmp; rm -rf /tmp/profr
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profr -- python3 $R/bench.py --no-cpu --evidence-launches 0 > $R/gpurun_out/rocprof_default.log 2>&1
find /tmp/profr -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/default_kernel_stats.csv \;
cd $R
tail -3 gpurun_out/pytest_gpu.log; tail -2 gpurun_out/smoke.log; grep "^{" gpurun_out/bench.log | cut -c1-3000; grep "^{" gpurun_out/bench_20.log | cut -c1-700; grep "^{" gpurun_out/bench_torchrun.log | cut -c1-300; head -3 gpurun_out/default_kernel_stats.csv | cut -c1-300
