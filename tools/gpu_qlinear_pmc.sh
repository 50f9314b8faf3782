#!/bin/bash
# LDS bank-conflict counters of the tiled integer consumer kernel (separate --pmc pass, kernel-trace only)
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
cd /tmp; rm -rf /tmp/pmcq
timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d /tmp/pmcq -- python3 $R/tools/qlinear_tiled_once.py > $R/gpurun_out/pmc_qlinear.log 2>&1
find /tmp/pmcq -name "*counter_collection.csv" -exec cp {} $R/gpurun_out/qlinear_tiled_lds_counters.csv \;
cd $R
python3 - <<'PY'
import csv, collections
acc = collections.defaultdict(float); n = collections.Counter()
try:
    for r in csv.DictReader(open('gpurun_out/qlinear_tiled_lds_counters.csv')):
        if 'qgemm' in r.get('Kernel_Name', ''):
            acc[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
    for k in acc: print(k, acc[k] / max(1, n[k]), 'per dispatch-row over', n[k], 'rows')
    if acc.get('SQ_LDS_IDX_ACTIVE'): print('conflict share of LDS-array cycles: %.4f' % (acc['SQ_LDS_BANK_CONFLICT'] / acc['SQ_LDS_IDX_ACTIVE']))
except Exception as e:
    print('parse failed', e); print(open('gpurun_out/pmc_qlinear.log').read()[-2000:])
PY
