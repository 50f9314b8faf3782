#!/bin/bash
mkdir -p gpurun_out/r04; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04
cd $R
timeout 600 python -m pytest tests/test_accelerate.py -m gpu -q > $O/pytest_accel.log 2>&1; tail -3 $O/pytest_accel.log
MCTQ_BINDING=ctypes timeout 1800 python -m pytest tests -m gpu -q > $O/pytest_gpu_ctypes.log 2>&1; echo "ctypes rc=$?" >> $O/pytest_gpu_ctypes.log; tail -4 $O/pytest_gpu_ctypes.log
MCTQ_ROCTX=1 timeout 1800 python -m pytest tests -m gpu -q -k "not every_float and not 2_32" > $O/pytest_gpu_roctx.log 2>&1; echo "roctx rc=$?" >> $O/pytest_gpu_roctx.log; tail -4 $O/pytest_gpu_roctx.log
