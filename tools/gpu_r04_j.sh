#!/bin/bash
# grid rounds on the flat (per-tensor) kernel: eager config 3 at batch sizes around one round of 2048 four-step blocks
mkdir -p gpurun_out/r04j; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04j
cd $R
for n in 40 48 52 55 56 57 60 64 72 80 96 111 112 113 128; do
  for u in 4 8 2; do
    timeout 200 python bench.py --config cfg3 --batch $n --stream-depth -1 --steps 400 --warmup 50 --prewarm-seconds 0.5 --no-cpu --no-eager-extra --evidence-launches 0 --unroll $u 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; n=$n; blocks=(n*150528//4 + 256*$u-1)//(256*$u)
print('N', n, 'unroll', $u, 'blocks', blocks, 'rounds', round(blocks/2048,2), '| us', round(r['kernel_us'],2), 'frac', round(r['frac'],3), r['kernel'])" | tee -a $O/flat_grid_rounds.log
  done
done
