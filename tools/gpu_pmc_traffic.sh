#!/bin/bash
# HBM traffic of the headline kernel: separate rocprofv3 --pmc passes (FETCH_SIZE costs 3 TCC slots, WRITE_SIZE 2),
# run on the GPU box via gpurun; results go to gpurun_out/ and are summarised into profiles/pmc_traffic.json
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
cd /tmp
for pass in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$pass
  timeout 600 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d /tmp/pmc_$pass -- python3 $R/bench.py --no-cpu --steps 100 --warmup 10 > $R/gpurun_out/pmc_$pass.log 2>&1
  f=$(find /tmp/pmc_$pass -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then head -1 $f > $R/gpurun_out/pmc_$pass.csv; grep rows_kernel $f | tail -40 >> $R/gpurun_out/pmc_$pass.csv; fi
done
head -3 $R/gpurun_out/pmc_FETCH_SIZE.csv; head -3 $R/gpurun_out/pmc_WRITE_SIZE.csv
