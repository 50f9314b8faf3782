#!/bin/bash
# refresh the per-config bench lines and rocprof summaries committed under profiles/
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
: > gpurun_out/bench_all_configs.jsonl
for c in cfg1 cfg2 cfg4 cfg5; do timeout 300 python bench.py --no-cpu --config $c --steps 200 >> gpurun_out/bench_all_configs.jsonl 2>/dev/null; done
for n in 1 8 64 256; do timeout 300 python bench.py --no-cpu --config cfg3 --batch $n --steps 300 >> gpurun_out/bench_all_configs.jsonl 2>/dev/null; done
timeout 300 python bench.py --no-cpu --extras >> gpurun_out/bench_all_configs.jsonl 2>/dev/null
timeout 300 python bench.py --no-cpu --streams 2 >> gpurun_out/bench_all_configs.jsonl 2>/dev/null
timeout 300 python bench.py --no-cpu --ring 1 >> gpurun_out/bench_all_configs.jsonl 2>/dev/null
cd /tmp
for c in cfg4 cfg5; do
  rm -rf /tmp/prof_$c; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$c -- python3 $R/bench.py --no-cpu --config $c --steps 100 > /dev/null 2>&1
  find /tmp/prof_$c -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/${c}_kernel_stats.csv \;
done
cd $R
python - <<'PY'
import json
for l in open('gpurun_out/bench_all_configs.jsonl'):
    try: d=json.loads(l)
    except Exception: continue
    c=d['config']
    print('%-52s %-20s streams=%d %s us=%7.2f GB/s=%5.0f frac=%.3f Gelem/s=%6.1f' % (c['workload'], c['shape'], c['streams'], c['cache_protocol'], d['roofline']['kernel_us'], d['achieved_gbs'], d['roofline']['frac'], d['value']/1e9))
PY
head -3 gpurun_out/cfg4_kernel_stats.csv; head -3 gpurun_out/cfg5_kernel_stats.csv
