#!/bin/bash
mkdir -p gpurun_out/r04h; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04h
cd $R
run() { local label=$1; shift
  timeout 300 python bench.py --no-cpu --evidence-launches 0 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$label', '| us', round(r['kernel_us'],2), 'frac', round(r['frac'],3), 'wall', round(r['frac_wall'],3), r['kernel'])" | tee -a $O/bf16_tuning.log
}
for rep in 1 2; do
run "bf16 4096^2 default" --dtype bf16
run "bf16 4096^2 nt=1" --dtype bf16 --nt 1
run "bf16 4096^2 rowsteps" --dtype bf16 --rowsteps 1
run "bf16 4096^2 rowsteps nt=1" --dtype bf16 --rowsteps 1 --nt 1
done
run "bf16 cfg5 default" --dtype bf16 --config cfg5 --steps 300
run "f32 cfg2 default" 
run "f32 cfg2 nt=2" --nt 2
