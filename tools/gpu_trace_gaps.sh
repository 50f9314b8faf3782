#!/bin/bash
# kernel time vs launch gaps (rocprofv3 --kernel-trace only) for the small-activation stream and the headline config
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
cd /tmp
: > $R/gpurun_out/trace_gaps.log
for spec in "cfg3_n1 --config cfg3 --batch 1 --steps 2000" "cfg3_n8 --config cfg3 --batch 8 --steps 2000" "cfg2 --steps 1000"; do
  set -- $spec; name=$1; shift
  rm -rf /tmp/tr_$name
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$name -- python3 $R/bench.py --no-cpu --evidence-launches 0 --prewarm-seconds 0.3 "$@" > $R/gpurun_out/trace_$name.log 2>&1
  f=$(find /tmp/tr_$name -name "*kernel_trace.csv" | head -1)
  python3 $R/tools/trace_gaps.py $f "$name (bench.py $*)" >> $R/gpurun_out/trace_gaps.log 2>&1
  tail -1 $R/gpurun_out/trace_$name.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('    bench line of the same run: ms_per_step %.5f  kernel_us(period by events) %.2f' % (d['ms_per_step'], d['roofline']['kernel_us']))" >> $R/gpurun_out/trace_gaps.log 2>&1
done
cat $R/gpurun_out/trace_gaps.log
