#!/bin/bash
# rocprofv3 kernel stats of the integer-consumer probe (exactness sweep + timings)
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
cd /tmp; rm -rf /tmp/profq
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profq -- python3 $R/tools/qlinear_probe.py > $R/gpurun_out/rocprof_qlinear.log 2>&1
find /tmp/profq -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/qlinear_kernel_stats.csv \;
cd $R; head -12 gpurun_out/qlinear_kernel_stats.csv | cut -c1-260
