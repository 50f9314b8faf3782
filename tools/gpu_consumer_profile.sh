#!/bin/bash
# rocprofv3 kernel stats of tools/bench_consumer.py (one line per problem size): the average duration of each consumer
# kernel must agree with the bench lines' HIP-event periods -> gpurun_out/r03/consumer_kernel_stats.csv
mkdir -p gpurun_out/r03; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
cd /tmp; rm -rf /tmp/profc
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profc -- python3 $R/tools/bench_consumer.py ${1:-unknown} > $R/gpurun_out/r03/bench_consumer_under_rocprof.jsonl 2> $R/gpurun_out/r03/rocprof_consumer.err
find /tmp/profc -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/r03/consumer_kernel_stats.csv \;
cd $R; grep -i "qlinear\|qgemm" gpurun_out/r03/consumer_kernel_stats.csv | cut -c1-200 | head -20
