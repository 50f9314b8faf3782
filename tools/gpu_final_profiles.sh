#!/bin/bash
# refresh the per-config bench lines committed under profiles/<round>/
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout 300 python -m pytest tests -m gpu -q -k "capture or fuzz" > gpurun_out/pytest_sel.log 2>&1; echo "rc=$?" >> gpurun_out/pytest_sel.log
: > gpurun_out/bench_all_configs.jsonl
for c in cfg1 cfg2 cfg4 cfg5; do timeout 300 python bench.py --no-cpu --config $c --steps 200 2>/dev/null | grep "^{" >> gpurun_out/bench_all_configs.jsonl; done
for n in 1 8 64 256; do timeout 300 python bench.py --no-cpu --config cfg3 --batch $n --steps 500 2>/dev/null | grep "^{" >> gpurun_out/bench_all_configs.jsonl; done
timeout 300 python bench.py --no-cpu --extras 2>/dev/null | grep "^{" >> gpurun_out/bench_all_configs.jsonl
timeout 300 python bench.py --no-cpu --streams 2 2>/dev/null | grep "^{" >> gpurun_out/bench_all_configs.jsonl
timeout 300 python bench.py --no-cpu --ring 1 2>/dev/null | grep "^{" >> gpurun_out/bench_all_configs.jsonl
timeout 300 python bench.py --no-cpu --graph --steps 200 2>/dev/null | grep "^{" >> gpurun_out/bench_all_configs.jsonl
timeout 300 python bench.py --no-cpu --config cfg3 --batch 1 --graph --steps 500 2>/dev/null | grep "^{" >> gpurun_out/bench_all_configs.jsonl
timeout 300 python bench.py --no-cpu --config cfg3 --batch 8 --graph --steps 500 2>/dev/null | grep "^{" >> gpurun_out/bench_all_configs.jsonl
tail -3 gpurun_out/pytest_sel.log
python3 - <<'PY'
import json
for l in open('gpurun_out/bench_all_configs.jsonl'):
    try: d=json.loads(l)
    except Exception: continue
    c=d['config']
    print('%-52s %-20s %-8s streams=%d %-4s period_us=%7.2f GB/s=%5.0f frac=%.3f Gelem/s=%6.1f traffic=%s' % (c['workload'], c['shape'], c['launch'], c['streams'], c['cache_protocol'], d['roofline']['kernel_us'], d['achieved_gbs'], d['roofline']['frac'], d['value']/1e9, d['roofline']['traffic']))
PY
