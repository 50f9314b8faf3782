#!/bin/bash
mkdir -p gpurun_out; : > gpurun_out/bench_lut.log
timeout 1800 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
for hu in 1 2 4; do timeout 300 python bench.py --no-cpu --config cfg4 --steps 200 --heavy-unroll $hu >> gpurun_out/bench_lut.log 2>&1; done
tail -3 gpurun_out/pytest_gpu.log
python - <<'PY'
import json
for l in open('gpurun_out/bench_lut.log'):
    try: d=json.loads(l)
    except Exception: continue
    print(d['config']['workload'], 'us=%.2f GB/s=%.0f frac=%.3f' % (d['roofline']['kernel_us'], d['achieved_gbs'], d['roofline']['frac']))
PY
