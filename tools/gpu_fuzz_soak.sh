#!/bin/bash
# Seeded fuzz suites on further seeds (MCTQ_FUZZ_SEED shifts every suite's generator), 300 cases each where the suite
# takes a case count; one summary line per seed -> gpurun_out/<round>/fuzz_soak.log.
mkdir -p gpurun_out/${MCTQ_ROUND:-r05}
: > gpurun_out/${MCTQ_ROUND:-r05}/fuzz_soak.log
for seed in ${SEEDS:-11 12 13 14 15 16}; do
  MCTQ_FUZZ_SEED=$seed MCTQ_FUZZ_CASES=${CASES:-300} python -m pytest tests -q -m gpu -k "fuzz" -x 2>&1 | tail -1 | sed "s/^/seed $seed: /" >> gpurun_out/${MCTQ_ROUND:-r05}/fuzz_soak.log
done
cat gpurun_out/${MCTQ_ROUND:-r05}/fuzz_soak.log
