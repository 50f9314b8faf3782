#!/bin/bash
# round 2, second GPU pass: full suite on the new code, launch-API probe, residency caps, e2e with the batcher, bench protocol
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
timeout 120 ./tools/launch_probe.bin > gpurun_out/launch_probe.log 2>&1; echo "rc=$?" >> gpurun_out/launch_probe.log
timeout 300 ./tools/kbench_timeline.bin 200 > gpurun_out/kbench_timeline.log 2>&1; echo "timeline rc=$?" >> gpurun_out/kbench_timeline.log
timeout 300 python tools/host_overhead.py > gpurun_out/host_overhead.log 2>&1; echo "rc=$?" >> gpurun_out/host_overhead.log
timeout 900 python tools/e2e_model.py > gpurun_out/e2e.log 2>&1; echo "rc=$?" >> gpurun_out/e2e.log
timeout 600 python bench.py --no-cpu > gpurun_out/bench_1000.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench_1000.log
timeout 600 python bench.py --no-cpu --steps 20 --warmup 5 > gpurun_out/bench_20.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench_20.log
timeout 600 python bench.py --no-cpu --steps 20 --warmup 5 --prewarm-seconds 0 > gpurun_out/bench_20_noprewarm.log 2>&1
for n in 1 8; do timeout 300 python bench.py --no-cpu --config cfg3 --batch $n --steps 2000 --warmup 100 > gpurun_out/bench_cfg3_n$n.log 2>&1; done
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --gather --steps 100 --warmup 10 --no-cpu > gpurun_out/bench_torchrun_gather.log 2>&1; echo "torchrun rc=$?" >> gpurun_out/bench_torchrun_gather.log
tail -6 gpurun_out/pytest_gpu.log; cat gpurun_out/launch_probe.log; sed -n '/\[D\]/,/\[C\]/p' gpurun_out/kbench_timeline.log; cat gpurun_out/host_overhead.log; cat gpurun_out/e2e.log | cut -c1-700
for f in bench_1000 bench_20 bench_20_noprewarm bench_cfg3_n1 bench_cfg3_n8 bench_torchrun_gather; do echo "== $f"; tail -2 gpurun_out/$f.log | cut -c1-2500; done
