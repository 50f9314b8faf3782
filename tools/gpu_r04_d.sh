#!/bin/bash
mkdir -p gpurun_out/r04d; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04d
cd $R
run() { # label, env, args
  local label=$1; shift; local envs=$1; shift
  env $envs timeout 300 python bench.py --config cfg4 --steps 300 --no-cpu --evidence-launches 0 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$label', 'us', round(r['kernel_us'],2), 'frac', round(r['frac'],3), 'wall', round(r['frac_wall'],3), r['kernel'])" | tee -a $O/cfg4_variants.log
}
run "table U=4 (r03 shipped)" MCTQ_COMPACT_LUT=0
run "table U=4 again" MCTQ_COMPACT_LUT=0
run "compact default (U=1)" MCTQ_COMPACT_LUT=1
run "compact U=4" MCTQ_COMPACT_LUT=1 --heavy-unroll 4
run "compact U=2" MCTQ_COMPACT_LUT=1 --heavy-unroll 2
run "compact persistent U=1" MCTQ_COMPACT_LUT=1 --heavy-persistent 1 --heavy-unroll 1
run "compact persistent U=2" MCTQ_COMPACT_LUT=1 --heavy-persistent 1 --heavy-unroll 2
run "compact persistent U=4" MCTQ_COMPACT_LUT=1 --heavy-persistent 1 --heavy-unroll 4
run "table persistent U=4" MCTQ_COMPACT_LUT=0 --heavy-persistent 1 --heavy-unroll 4
run "table persistent U=1" MCTQ_COMPACT_LUT=0 --heavy-persistent 1 --heavy-unroll 1
run "compact default again" MCTQ_COMPACT_LUT=1
