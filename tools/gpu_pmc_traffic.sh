#!/bin/bash
# Per-config profiler evidence for the judged bench command, run on the GPU box via gpurun:
#   1. rocprofv3 --kernel-trace --stats of `bench.py --config C` (kernel durations; no counters)
#   2. separate rocprofv3 --kernel-trace --pmc passes (never combined with other trace domains):
#        FETCH_SIZE, WRITE_SIZE                      HBM-side traffic (FETCH_SIZE costs 3 TCC slots, WRITE_SIZE 2)
#        SQ stall split, TCC EA stall counters       who waits for what (cfg2 only)
# Raw per-dispatch CSVs land in gpurun_out/pmc/<config>/; tools/pmc_summarize.py (run in the build container
# afterwards, MCTQ_ROUND=r04) turns them into profiles/pmc_traffic.json + profiles/<round>/*.csv, keyed by the kernel VARIANT that the
# very same bench run reports (mctq_last_launch) and the git head.
mkdir -p gpurun_out/pmc; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 -L 2>/dev/null | grep -o -E "\b(SQ_[A-Z_0-9]+|TCC_[A-Z_0-9]+|TCP_[A-Z_0-9]+|GRBM_[A-Z_0-9]+|FETCH_SIZE|WRITE_SIZE)\b" | sort -u > $R/gpurun_out/pmc/available_counters.txt
run_cfg() {   # name, bench args...   (MCTQ_PMC_ONLY="cfg2_bf16 cfg2_f16": only those configurations)
  local name=$1; shift
  if [ -n "$MCTQ_PMC_ONLY" ] && ! echo " $MCTQ_PMC_ONLY " | grep -q " $name "; then return; fi
  local out=$R/gpurun_out/pmc/$name; mkdir -p $out
  rm -rf /tmp/prof_$name
  # the stats pass runs the judged command as it is (at N = 1 the default run also carries the batched_16x4096 object)
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -- python3 $R/bench.py --no-cpu --no-eager-extra --no-sharded-extra --evidence-launches 0 "$@" > $out/bench_stats.log 2>&1
  find /tmp/prof_$name -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats.csv \;
  for pass in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_${name}_$pass
    timeout 600 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d /tmp/pmc_${name}_$pass -- python3 $R/bench.py --no-cpu --no-batched-extra --no-eager-extra --no-sharded-extra --evidence-launches 0 --prewarm-seconds 0.2 --steps 100 --warmup 10 "$@" > $out/bench_$pass.log 2>&1
    f=$(find /tmp/pmc_${name}_$pass -name "*counter_collection.csv" | head -1)
    if [ -n "$f" ]; then head -1 $f > $out/$pass.csv; grep -E "mctq" $f | tail -120 >> $out/$pass.csv; fi
  done
}
run_cfg cfg2
run_cfg cfg2_batched16 --batched 16 --steps 60 --warmup 5
run_cfg resnet50 --config resnet50 --steps 200 --warmup 10
run_cfg cfg3_n64 --config cfg3 --batch 64
run_cfg cfg3_n64_eager --config cfg3 --batch 64 --stream-depth -1
run_cfg cfg3_n8 --config cfg3 --batch 8
# per-tensor launches of 3/4 ... 1 round of resident blocks (flat_paced_kernel, round 6): 50 x 3 x 224 x 224 float32, 100 x ... bfloat16
run_cfg cfg3_n50_eager --config cfg3 --batch 50 --stream-depth -1
run_cfg cfg3_n100_eager_bf16 --config cfg3 --batch 100 --dtype bf16 --stream-depth -1
run_cfg sym_4096x4096_pertensor_bf16 --config sym --shape 4096x4096 --per-tensor --dtype bf16
run_cfg sym_2048x4096_pertensor --config sym --shape 2048x4096 --per-tensor
run_cfg sym_2048x4096_axis0 --config sym --shape 2048x4096 --axis 0
run_cfg cfg4 --config cfg4 --steps 300
run_cfg cfg5 --config cfg5 --steps 300
# 16-bit storage: the headline shape and config 5 as bfloat16, config 4 (LUT: 2 B in, 4 B out)
run_cfg cfg2_bf16 --dtype bf16
run_cfg cfg2_f16 --dtype f16
run_cfg cfg5_bf16 --config cfg5 --dtype bf16 --steps 300
run_cfg cfg4_bf16 --config cfg4 --dtype bf16 --steps 300
# round 6: the launch shapes VERDICT r05 #1 names (16-bit channel-last and short / ragged rows), and their float32 twins
run_cfg sym_4096x4096_axis1_bf16 --config sym --shape 4096x4096 --axis 1 --dtype bf16
run_cfg sym_65536x200_axis1_bf16 --config sym --shape 65536x200 --axis 1 --dtype bf16
run_cfg sym_16384x1020_axis0_bf16 --config sym --shape 16384x1020 --axis 0 --dtype bf16
run_cfg sym_1048576x16_axis0_bf16 --config sym --shape 1048576x16 --axis 0 --dtype bf16
run_cfg sym_16384x1024_axis0_bf16 --config sym --shape 16384x1024 --axis 0 --dtype bf16
run_cfg sym_64x56x56x256_axis3_bf16 --config sym --shape 64x56x56x256 --axis 3 --dtype bf16
run_cfg sym_4096x4096_axis1 --config sym --shape 4096x4096 --axis 1
# stall counters of the headline kernel and of the batched launch: one pass per group (8 SQ slots, 4 TCC slots)
stall_passes() {   # name, bench args...
  local name=$1; shift
  if [ -n "$MCTQ_PMC_ONLY" ] && ! echo " $MCTQ_PMC_ONLY " | grep -q " $name "; then return; fi
  local out=$R/gpurun_out/pmc/$name
  local i=0
  for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
             "SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM GRBM_GUI_ACTIVE" \
             "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum" \
             "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
             "TCC_EA0_RD_UNCACHED_32B_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_64B_sum" \
             "TCC_TAG_STALL_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum" \
             "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_NC_READ_REQ_sum"; do
    i=$((i+1)); rm -rf /tmp/pmc_stall_${name}_$i
    timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pmc_stall_${name}_$i -- python3 $R/bench.py --no-cpu --no-batched-extra --no-sharded-extra --evidence-launches 0 --prewarm-seconds 0.2 --steps 60 --warmup 10 "$@" > $out/bench_stall_$i.log 2>&1
    f=$(find /tmp/pmc_stall_${name}_$i -name "*counter_collection.csv" | head -1)
    if [ -n "$f" ]; then head -1 $f > $out/stall_$i.csv; grep -E "mctq" $f | tail -400 >> $out/stall_$i.csv; else echo "$name group $i: no counter file (unknown counter name?)"; tail -3 $out/bench_stall_$i.log; fi
  done
}
stall_passes cfg2
stall_passes cfg2_batched16 --batched 16 --steps 30 --warmup 5
stall_passes cfg2_bf16 --dtype bf16
stall_passes sym_4096x4096_axis1_bf16 --config sym --shape 4096x4096 --axis 1 --dtype bf16
stall_passes sym_65536x200_axis1_bf16 --config sym --shape 65536x200 --axis 1 --dtype bf16
stall_passes sym_16384x1020_axis0_bf16 --config sym --shape 16384x1020 --axis 0 --dtype bf16
stall_passes sym_1048576x16_axis0_bf16 --config sym --shape 1048576x16 --axis 0 --dtype bf16
ls -la $R/gpurun_out/pmc/*; wc -l $R/gpurun_out/pmc/available_counters.txt
for c in cfg2 cfg2_batched16 resnet50 cfg3_n64 cfg3_n64_eager cfg3_n8 cfg3_n50_eager cfg3_n100_eager_bf16 sym_4096x4096_pertensor_bf16 sym_2048x4096_pertensor sym_2048x4096_axis0 cfg4 cfg5 cfg2_bf16 cfg2_f16 cfg5_bf16 cfg4_bf16 sym_4096x4096_axis1_bf16 sym_65536x200_axis1_bf16 sym_16384x1020_axis0_bf16 sym_1048576x16_axis0_bf16 sym_16384x1024_axis0_bf16 sym_64x56x56x256_axis3_bf16 sym_4096x4096_axis1; do head -2 $R/gpurun_out/pmc/$c/kernel_stats.csv | cut -c1-260; tail -1 $R/gpurun_out/pmc/$c/bench_stats.log | cut -c1-200; done
