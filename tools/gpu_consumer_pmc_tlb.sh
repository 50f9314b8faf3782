#!/bin/bash
mkdir -p gpurun_out/r03; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03/consumer_pmc_m256_tlb.csv; echo "variant,counter,mean_per_dispatch,dispatches" > $OUT
cd /tmp
for v in 662 166623; do
  for pass in "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum" \
              "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_ACCESSES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum" \
              "TA_BUSY_avr TA_ADDR_FIFO_FULL_sum TA_DATA_FIFO_FULL_sum TA_CMD_FIFO_FULL_sum" \
              "TCP_TCP_TA_ADDR_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum TCP_TCR_RDRET_STALL_sum"; do
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
    v = v[len(v) // 2:]
    print(f"{sys.argv[2]},{k},{sum(v) / len(v):.1f},{len(v)}")
PY
    [ -z "$f" ] && tail -3 /tmp/pmcc.log
  done
done
cd $R; cat $OUT
