#!/bin/bash
# Round 6: the launch-geometry fuzz (tests/test_gpu_affine.py) on further seeds, the implicit-wrap bench test -> gpurun_out/r06/
O=gpurun_out/r06; mkdir -p $O
python - <<'PY' > $O/build_id.txt 2>&1
from mct_quantizers_amd.hip import native
print(native.load().mctq_build_id().decode())
PY
: > $O/fuzz_geometry.log
for seed in ${SEEDS:-606 61 62 63 64 65 66 67 68 69 70 71}; do
  MCTQ_FUZZ_SEED=$seed MCTQ_FUZZ_CASES=${CASES:-3000} timeout 900 python -m pytest tests/test_gpu_affine.py -q -m gpu -k "fuzz_channel_last" -x 2>&1 | tail -4 | sed "s/^/seed $seed: /" >> $O/fuzz_geometry.log
done
timeout 900 python -m pytest tests/test_gpu_multigpu.py -q -m gpu -k "beyond_the_visible" -x > $O/pytest_implicit_wrap.log 2>&1; echo "rc=$?" >> $O/pytest_implicit_wrap.log
cat $O/build_id.txt $O/fuzz_geometry.log; tail -5 $O/pytest_implicit_wrap.log
