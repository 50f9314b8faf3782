#!/bin/bash
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
cd /tmp; rm -rf /tmp/roctx_prof
MCTQ_ROCTX=1 timeout 600 rocprofv3 --kernel-trace --marker-trace --stats --output-format csv -d /tmp/roctx_prof -- python3 $R/bench.py --no-cpu --steps 50 --warmup 5 > $R/gpurun_out/roctx.log 2>&1
find /tmp/roctx_prof -name "*marker*stats*.csv" -exec head -5 {} \;
find /tmp/roctx_prof -name "*marker_api_trace.csv" -exec sh -c 'head -3 {}; wc -l {}' \;
cd $R; python -m pytest tests -m gpu -q -x -k "golden_cases_via or side_stream" 2>&1 | tail -2
python tools/host_overhead.py 2>&1 | grep "quantizer(x)"
