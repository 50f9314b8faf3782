#!/bin/bash
# first GPU contact: tests, smoke, bench variants, kernel trace
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import torch; print(torch.cuda.get_device_name(0))" > gpurun_out/dev.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
for nt in 0 1; do for u in 1 2 4 8; do
  timeout 300 python bench.py --no-cpu --nt $nt --unroll $u --steps 400 >> gpurun_out/bench_variants.log 2>&1
done; done
timeout 300 python bench.py --no-cpu --graph --steps 400 >> gpurun_out/bench_variants.log 2>&1
timeout 300 python bench.py --no-cpu --ring 1 --steps 400 >> gpurun_out/bench_variants.log 2>&1
timeout 600 python bench.py > gpurun_out/bench.log 2>&1
tail -3 gpurun_out/pytest_gpu.log; cat gpurun_out/smoke.log | tail -3; cat gpurun_out/bench_variants.log | python -c "
import sys, json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: print(l.strip()[:200]); continue
    print(d['config']['launch'], d['config']['cache_protocol'], 'us/launch=%.2f GB/s=%.0f frac=%.3f ms/step=%.4f' % (d['roofline']['kernel_us'], d['achieved_gbs'], d['roofline']['frac'], d['ms_per_step']))
"
tail -1 gpurun_out/bench.log
