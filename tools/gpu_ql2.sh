#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_consumers.py -m gpu -q > gpurun_out/pytest_consumers.log 2>&1; echo "rc=$?" >> gpurun_out/pytest_consumers.log
timeout 600 python tools/bench_consumer.py > gpurun_out/bench_consumer.jsonl 2>gpurun_out/bench_consumer.err
timeout 900 python tools/e2e_model.py 2>&1 | grep -v Adjusting > gpurun_out/e2e.log
tail -3 gpurun_out/pytest_consumers.log; python3 - <<'PY'
import json
for l in open('gpurun_out/bench_consumer.jsonl'):
    d=json.loads(l); print(d['config']['workload'], 'us=%.2f' % (d['ms_per_step']*1e3), d['roofline']['bound'], 'frac=%.3f' % d['roofline']['frac'])
PY
cut -c1-700 gpurun_out/e2e.log
