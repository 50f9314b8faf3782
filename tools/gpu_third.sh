#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
: > gpurun_out/bench_cfgs.log
for c in cfg1 cfg2 cfg4 cfg5; do timeout 300 python bench.py --no-cpu --config $c --steps 200 >> gpurun_out/bench_cfgs.log 2>&1; done
for n in 1 8 64 256; do timeout 300 python bench.py --no-cpu --config cfg3 --batch $n --steps 300 >> gpurun_out/bench_cfgs.log 2>&1; done
cd /tmp
for pass in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$pass
  timeout 600 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d /tmp/pmc_$pass -- python3 $R/bench.py --no-cpu --steps 100 --warmup 10 > $R/gpurun_out/pmc_$pass.log 2>&1
  f=$(find /tmp/pmc_$pass -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then head -1 $f > $R/gpurun_out/pmc_$pass.csv; grep rows_kernel $f | tail -40 >> $R/gpurun_out/pmc_$pass.csv; fi
done
rm -rf /tmp/prof4; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof4 -- python3 $R/bench.py --no-cpu --config cfg4 --steps 100 > $R/gpurun_out/rocprof_cfg4.log 2>&1
find /tmp/prof4 -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/cfg4_kernel_stats.csv \;
cd $R
python - <<'PY'
import json
for l in open('gpurun_out/bench_cfgs.log'):
    try: d=json.loads(l)
    except Exception: continue
    print(d['config']['workload'], d['config']['shape'], 'us=%.2f GB/s=%.0f frac=%.3f Gelem/s=%.1f ring=%d' % (d['roofline']['kernel_us'], d['achieved_gbs'], d['roofline']['frac'], d['value']/1e9, d['config']['buffer_ring']))
PY
head -3 gpurun_out/pmc_FETCH_SIZE.csv; head -3 gpurun_out/pmc_WRITE_SIZE.csv; cat gpurun_out/cfg4_kernel_stats.csv | head -5
