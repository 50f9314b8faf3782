#!/bin/bash
mkdir -p gpurun_out/r04d; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04d
cd $R
for v in "compact0 MCTQ_COMPACT_LUT=0 0" "compact1_U1 MCTQ_COMPACT_LUT=1 1" "compact1_U4 MCTQ_COMPACT_LUT=1 4" "table_warm_out MCTQ_PROBE_WARM_OUT=1 0"; do
  set -- $v
  echo "== $1" | tee -a $O/lut_vs_affine_cold.log
  env $2 MCTQ_COMPACT_LUT=${2#MCTQ_COMPACT_LUT=} timeout 200 python tools/lut_vs_affine.py 150 $3 2>&1 | grep -v amdgpu.ids | tee -a $O/lut_vs_affine_cold.log
done
for v in STAGE1 STAGE4 STAGE1_DIV_LDS; do for hu in 0 1; do
  echo "== ablation $v heavy_unroll=$hu (cold outputs)" | tee -a $O/lut_vs_affine_cold.log
  MCTQ_COMPACT_LUT=0 MCTQ_HIP_LIB=$R/tools/ablate/libmctq_hip_$v.so MCTQ_BINDING=ctypes timeout 200 python tools/lut_vs_affine.py 150 $hu 2>&1 | grep -v amdgpu.ids | grep lut | tee -a $O/lut_vs_affine_cold.log
done; done
