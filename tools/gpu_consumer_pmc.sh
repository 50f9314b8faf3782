#!/bin/bash
# Counters of the consumer's tiled kernel at M = 256 (4096 x 4096 weights, cold) with 4 / 8 / 16 waves per block:
# separate --pmc passes with --kernel-trace only -> gpurun_out/r03/consumer_pmc_m256.csv (one row per variant x counter)
mkdir -p gpurun_out/r03; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03/consumer_pmc_m256.csv; echo "variant,counter,mean_per_dispatch,dispatches" > $OUT
cd /tmp
for v in 662 6623 86623 166623; do
  for pass in "GRBM_GUI_ACTIVE TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum" \
              "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
              "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" \
              "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TA_ADDR_FIFO_FULL_sum"; do
    rm -rf /tmp/pmcc
    timeout 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d /tmp/pmcc -- python3 $R/tools/qlinear_variant_once.py 256 $v > /tmp/pmcc.log 2>&1
    f=$(find /tmp/pmcc -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && python3 - "$f" $v >> $OUT <<'PY'
import csv, collections, sys
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if 'qgemm' in r.get('Kernel_Name', ''):
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in acc.items():
    v = v[len(v) // 2:]                      # the later (cold-ring) dispatches
    print(f"{sys.argv[2]},{k},{sum(v) / len(v):.1f},{len(v)}")
PY
  done
done
cd $R; cat $OUT
