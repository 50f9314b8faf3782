#!/bin/bash
# round 4, closing trip: full GPU suite at the final head (default routing, then the LUT suites with the compact table), fuzz soak
mkdir -p gpurun_out/r04; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04
cd $R
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "all gpu tests rc=$?" >> $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
MCTQ_COMPACT_LUT=1 timeout 1200 python -m pytest tests -m gpu -q -k "lut or Lut or LUT or golden or stream or accelerate" > $O/pytest_gpu_compact_lut.log 2>&1; echo "compact rc=$?" >> $O/pytest_gpu_compact_lut.log; tail -3 $O/pytest_gpu_compact_lut.log
MCTQ_BINDING=ctypes timeout 1800 python -m pytest tests -m gpu -q -x --deselect tests/test_holder_fast_call.py > $O/pytest_gpu_ctypes.log 2>&1; echo "ctypes rc=$?" >> $O/pytest_gpu_ctypes.log; tail -3 $O/pytest_gpu_ctypes.log
SEEDS="21 22 23" bash tools/gpu_fuzz_soak.sh
MCTQ_COMPACT_LUT=1 SEEDS="24 25" bash tools/gpu_fuzz_soak.sh | sed 's/^/compact: /' | tee -a $O/fuzz_soak_compact.log
