#!/bin/bash
# SQ / LDS / wait counters of rows_kernel<LutTableOp> beside rows_kernel<AffineOp> on the config-4 tensor: separate
# rocprofv3 --kernel-trace --pmc passes (never combined with other trace domains), averaged per kernel by
# tools/pmc_kernel_table.py into profiles/r04/cfg4_lut_vs_affine_counters.csv
mkdir -p gpurun_out/lutpmc; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
cd /tmp
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM GRBM_GUI_ACTIVE" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_THREAD_CYCLES_VALU" \
           "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_IFETCH SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_INSTS" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum"; do
  i=$((i+1)); rm -rf /tmp/lutpmc_$i
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/lutpmc_$i -- python3 $R/tools/lut_vs_affine.py 40 > $R/gpurun_out/lutpmc/run_$i.log 2>&1
  f=$(find /tmp/lutpmc_$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then head -1 $f > $R/gpurun_out/lutpmc/group_$i.csv; grep -E "rows_kernel" $f >> $R/gpurun_out/lutpmc/group_$i.csv; else echo "group $i: no counter file"; tail -3 $R/gpurun_out/lutpmc/run_$i.log; fi
done
rm -rf /tmp/lutstats
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lutstats -- python3 $R/tools/lut_vs_affine.py 200 > $R/gpurun_out/lutpmc/run_stats.log 2>&1
find /tmp/lutstats -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/lutpmc/kernel_stats.csv \;
cat $R/gpurun_out/lutpmc/run_stats.log | tail -5
