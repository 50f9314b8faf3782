#!/bin/bash
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
: > gpurun_out/bench_cfgs2.log
for c in cfg2 cfg4 cfg5; do timeout 300 python bench.py --no-cpu --config $c --steps 200 >> gpurun_out/bench_cfgs2.log 2>&1; done
tail -4 gpurun_out/pytest_gpu.log
python - <<'PY'
import json
for l in open('gpurun_out/bench_cfgs2.log'):
    try: d=json.loads(l)
    except Exception: continue
    print(d['config']['workload'], d['config']['shape'], 'us=%.2f GB/s=%.0f frac=%.3f Gelem/s=%.1f ring=%d' % (d['roofline']['kernel_us'], d['achieved_gbs'], d['roofline']['frac'], d['value']/1e9, d['config']['buffer_ring']))
PY
