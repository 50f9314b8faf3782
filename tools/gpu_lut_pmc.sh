#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
cd /tmp; rm -rf /tmp/pmcl
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d /tmp/pmcl -- python3 $R/bench.py --no-cpu --config cfg4 --steps 20 --warmup 5 > $R/gpurun_out/pmc_lut.log 2>&1
f=$(find /tmp/pmcl -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
acc=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if 'rows_persist' in r['Kernel_Name']:
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in acc.items(): print(k, sum(v)/len(v), len(v))
PY
