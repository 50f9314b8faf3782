#!/bin/bash
# How round 5's logs under profiles/r05/ were produced on the MI355X box: `gpurun -- 'bash tools/gpu_r05.sh <section> ...'`.
# Everything is written under gpurun_out/r05/ and copied to profiles/r05/ by hand.  rocprofv3 --pmc passes never share a
# run with other trace domains; the profiled program follows `--` directly (python3 <script>).
mkdir -p gpurun_out/r05; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05
cd $R
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', '| us', round(r['kernel_us'],2), 'frac', round(r['frac'],3), 'wall', round(r['frac_wall'],3), r['kernel'])"; }

hist() {          # per-dispatch durations of the judged launch vs ring slot / output address / gap / time
  for v in "cfg2 f32 1.0" "cfg2 bf16 0.5" "cfg4 bf16 2.0"; do
    set -- $v; name=$1_$2
    rm -rf /tmp/tr_$name
    timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$name -- python3 $R/tools/dispatch_hist.py run --config $1 --dtype $2 --launches 3000 --log /tmp/tr_$name.json > $O/dispatch_${name}_run.log 2>&1
    timeout 300 python tools/dispatch_hist.py analyze --log /tmp/tr_$name.json --trace /tmp/tr_$name --out $O/${name}_dispatch --scale $3 > $O/dispatch_${name}_phases.log 2>&1
  done
}

launchlog() {     # every launch variant the GPU suite, the bench configurations and the shape sweeps select
  export MCTQ_LAUNCH_LOG=$O/launch_variants_raw.log; : > $MCTQ_LAUNCH_LOG
  timeout 1800 python -m pytest tests -m gpu -q -x > $O/pytest_gpu_launchlog.log 2>&1; echo "rc=$?" >> $O/pytest_gpu_launchlog.log
  cp $MCTQ_LAUNCH_LOG $O/launch_variants_suite.log
  for c in cfg2 cfg4 cfg5 resnet50 linear16; do for dt in f32 bf16 f16; do
    timeout 300 python bench.py --config $c --dtype $dt --steps 50 --warmup 5 --prewarm-seconds 0.2 --no-cpu --evidence-launches 0 > /dev/null 2>&1
  done; done
  for n in 1 8 64 256; do timeout 300 python bench.py --config cfg3 --batch $n --steps 50 --warmup 5 --prewarm-seconds 0.2 --no-cpu --evidence-launches 0 > /dev/null 2>&1; done
  timeout 300 python bench.py --config resnet50 --e2e --steps 20 > /dev/null 2>&1
  timeout 300 python bench.py --config resnet50 --e2e --e2e-lut --steps 20 > /dev/null 2>&1
  timeout 600 python tools/sweep_shapes.py > $O/sweep_shapes.log 2>&1
  timeout 600 python tools/short_rows_probe.py > /dev/null 2>&1
  timeout 600 python tools/bench_consumer.py launchlog > /dev/null 2>&1
  python __graft_entry__.py smoke > /dev/null 2>&1
  sort -u $MCTQ_LAUNCH_LOG > $O/launch_variants_all.log
  unset MCTQ_LAUNCH_LOG
  wc -l $O/launch_variants_suite.log $O/launch_variants_all.log
}

suite() {         # the GPU suite + the judged lines at the current head
  timeout 1800 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log
  tail -3 $O/pytest_gpu.log
}

bench() {         # judged line + side lines
  python __graft_entry__.py smoke > $O/smoke.log 2>&1
  timeout 300 python bench.py 2>/dev/null | tail -1 > $O/bench_default.json
  timeout 300 python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_20.json
  : > $O/bench_other_configs.jsonl; : > $O/bench_dtype.jsonl; : > $O/bench_cfg3.jsonl
  for c in cfg4 cfg5 resnet50; do timeout 300 python bench.py --config $c --steps 300 2>/dev/null | tail -1 >> $O/bench_other_configs.jsonl; done
  for dt in bf16 f16; do timeout 300 python bench.py --dtype $dt 2>/dev/null | tail -1 >> $O/bench_dtype.jsonl; done
  timeout 300 python bench.py --dtype bf16 --config cfg5 --steps 300 2>/dev/null | tail -1 >> $O/bench_dtype.jsonl
  timeout 300 python bench.py --dtype bf16 --config cfg4 --steps 300 2>/dev/null | tail -1 >> $O/bench_dtype.jsonl
  for n in 1 8 64; do timeout 300 python bench.py --config cfg3 --batch $n --steps 1000 --warmup 100 2>/dev/null | tail -1 >> $O/bench_cfg3.jsonl; done
  for f in bench_default.json bench_20.json; do line $f < $O/$f; done
  while read -r l; do echo "$l" | line other; done < $O/bench_other_configs.jsonl
  while read -r l; do echo "$l" | line dtype; done < $O/bench_dtype.jsonl
}

rehearse8() {     # the N = 8 entry path on the one GPU (8 ranks share it), RCCL refused -> needs --allow-gloo
  ( time MCTQ_BENCH_WRAP_DEVICES=1 timeout 1200 python bench.py --gpus 8 --steps 20 --warmup 5 --allow-gloo ) > $O/bench_gpus8_wrapped.log 2>&1
  echo "rc=$?" >> $O/bench_gpus8_wrapped.log
  ( time MCTQ_BENCH_WRAP_DEVICES=1 timeout 600 python bench.py --gpus 2 --steps 20 --warmup 5 ) > $O/bench_gpus2_wrapped_no_allow.log 2>&1
  echo "rc=$? (expected non-zero: two ranks on one GPU cannot form an RCCL group, and --allow-gloo was not given)" >> $O/bench_gpus2_wrapped_no_allow.log
  tail -5 $O/bench_gpus8_wrapped.log | cut -c1-600
}

placement() {     # VERDICT r04 #2, the one experiment: where input and output live vs the launch's duration
  timeout 900 python tools/placement_probe.py 2>&1 | grep -v amdgpu.ids > $O/placement_probe.log
  tail -50 $O/placement_probe.log
}

newtests() {      # the tests this round added, verbose (the versioned check prints its host cost)
  timeout 1800 python -m pytest tests/test_accelerate.py tests/test_gpu_multigpu.py tests/test_holder_fast_call.py tests/test_gpu_batched.py tests/test_gpu_streams.py -m gpu -q -s -x 2>&1 | grep -v amdgpu.ids | tail -40 > $O/pytest_new_tests.log
  tail -15 $O/pytest_new_tests.log
}

lutconf() {       # LDS layouts of the decision table (experiment library: python tools/build_variant.py lut_conflicts)
  timeout 600 python tools/experiments/lut_conflicts/run.py 2>&1 | grep -v amdgpu.ids > $O/lut_conflicts_timing.log
  cat $O/lut_conflicts_timing.log
  rm -rf /tmp/lutconf; mkdir -p $O/lutconf
  i=0
  for grp in "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_LDS" \
             "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU"; do
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/lutconf/$i -- python3 $R/tools/experiments/lut_conflicts/run.py pmc > $O/lutconf/run_$i.log 2>&1
    f=$(find /tmp/lutconf/$i -name "*counter_collection.csv" | head -1)
    if [ -n "$f" ]; then head -1 $f > $O/lutconf/group_$i.csv; grep -E "LutTableXOp" $f >> $O/lutconf/group_$i.csv; else echo "group $i: no counter file"; tail -3 $O/lutconf/run_$i.log; fi
  done
  python - <<'PYEOF'
import csv, glob, os, re
O = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r05")
acc = {}
for path in sorted(glob.glob(os.path.join(O, "lutconf", "group_*.csv"))):
    for r in csv.DictReader(open(path)):
        m = re.search(r"LutTableXOp<(\d)>", r["Kernel_Name"])
        if m:
            acc.setdefault(r["Counter_Name"], {}).setdefault(int(m.group(1)), []).append(float(r["Counter_Value"]))
with open(os.path.join(O, "lut_conflicts_counters.csv"), "w") as f:
    f.write("counter,mode0_shipped,mode1_swizzle,mode2_planes,mode3_one_cell,mode4_T_only\n")
    for c in sorted(acc):
        f.write(c + "," + ",".join(f"{sum(acc[c].get(m, [0])) / max(1, len(acc[c].get(m, []))):.0f}" for m in range(5)) + "\n")
print(open(os.path.join(O, "lut_conflicts_counters.csv")).read())
PYEOF
}

rowsched() {      # the rowsteps kernel with a written schedule (experiment library: python tools/build_variant.py rowsteps_sched)
  timeout 900 python tools/experiments/rowsteps_sched/run.py 2>&1 | grep -v amdgpu.ids > $O/rowsteps_sched.log
  cat $O/rowsteps_sched.log
  timeout 900 python tools/rowsteps_probe.py 2>&1 | grep -v amdgpu.ids > $O/rowsteps_probe_sustained.log     # 30 shapes x 3 storage types
  tail -5 $O/rowsteps_probe_sustained.log
}

pmc() {           # rocprof stats + FETCH / WRITE passes of every bench configuration -> gpurun_out/pmc (then: MCTQ_ROUND=r05 python tools/pmc_summarize.py)
  bash tools/gpu_pmc_traffic.sh > $O/pmc_traffic_run.log 2>&1
  tail -30 $O/pmc_traffic_run.log | cut -c1-220
}

modes() {         # the GPU suite through the other binding / with roctx ranges, and the seeded fuzz suites on further seeds
  MCTQ_BINDING=ctypes timeout 1800 python -m pytest tests -m gpu -q > $O/pytest_gpu_ctypes.log 2>&1; echo "rc=$?" >> $O/pytest_gpu_ctypes.log; tail -2 $O/pytest_gpu_ctypes.log
  MCTQ_ROCTX=1 timeout 1800 python -m pytest tests -m gpu -q -k "not every_float and not 2_32" > $O/pytest_gpu_roctx.log 2>&1; echo "rc=$?" >> $O/pytest_gpu_roctx.log; tail -2 $O/pytest_gpu_roctx.log
  MCTQ_ROUND=r05 SEEDS="31 32 33 34" bash tools/gpu_fuzz_soak.sh
}

e2e() {           # bench.py --config resnet50 --e2e lines on this round's library (AutoCapture now checks hooks / autocast / strides per call)
  for v in "resnet50" "resnet50_lut --e2e-lut" "resnet50_64px --e2e-side 64" "resnet50_b32 --batch 32 --steps 50"; do
    set -- $v; name=$1; shift
    timeout 600 python bench.py --config resnet50 --e2e --steps 100 "$@" 2>>$O/e2e.err | tail -1 > $O/bench_e2e_$name.json
    python -c "
import json; d=json.load(open('$O/bench_e2e_$name.json')); print('$name', {k: round(v, 3) if isinstance(v, float) else v for k, v in d.items() if k in ('value', 'ms_per_step') or k.startswith('ms_')}, {k: d[k] for k in d if 'ms' in k and k not in ('ms_per_step',)} if False else '')" 2>/dev/null || head -c 600 $O/bench_e2e_$name.json
  done
}

soak() {          # closing soak at the final head: the whole GPU suite three times in a row, four more fuzz seeds, the judged command three times
  : > $O/closing_soak.log
  for i in 1 2 3; do timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -1 | sed "s/^/suite run $i: /" >> $O/closing_soak.log; done
  MCTQ_ROUND=r05 SEEDS="41 42 43 44" bash tools/gpu_fuzz_soak.sh > /dev/null 2>&1; sed "s/^/fuzz /" $O/fuzz_soak.log >> $O/closing_soak.log
  for i in 1 2 3; do timeout 300 python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 | line "bench --steps 20 run $i" >> $O/closing_soak.log; done
  cat $O/closing_soak.log
}

lastaxis() {      # counters of the channel-last kernel beside the row kernels on the same tensor
  timeout 300 python tools/lastaxis_vs_rows.py 2>&1 | grep -v amdgpu.ids > $O/lastaxis_vs_rows.log; cat $O/lastaxis_vs_rows.log
  rm -rf /tmp/lax; mkdir -p $O/lax; i=0
  for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
             "SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM GRBM_GUI_ACTIVE SQ_LEVEL_WAVES" \
             "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
             "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum" \
             "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_RDREQ_32B_sum" \
             "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/lax/$i -- python3 $R/tools/lastaxis_vs_rows.py pmc > $O/lax/run_$i.log 2>&1
    f=$(find /tmp/lax/$i -name "*counter_collection.csv" | head -1)
    if [ -n "$f" ]; then head -1 $f > $O/lax/group_$i.csv; grep -E "mctq" $f >> $O/lax/group_$i.csv; else echo "group $i: no counter file"; tail -3 $O/lax/run_$i.log; fi
  done
  python - <<'PYEOF'
import csv, glob, os, re
O = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r05")
acc = {}
def short(n):
    m = re.search(r"(rows_kernel|rowsteps_kernel|lastaxis_kernel)", n)
    t = "bf16" if "DF16b" in n else "f32"
    return f"{m.group(1)}/{t}" if m else None
for path in sorted(glob.glob(os.path.join(O, "lax", "group_*.csv"))):
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"])
        if k:
            acc.setdefault(r["Counter_Name"], {}).setdefault(k, []).append(float(r["Counter_Value"]))
ks = sorted({k for v in acc.values() for k in v})
with open(os.path.join(O, "lastaxis_vs_rows_counters.csv"), "w") as f:
    f.write("counter," + ",".join(ks) + "\n")
    for c in sorted(acc):
        f.write(c + "," + ",".join(f"{sum(acc[c].get(k, [0])) / max(1, len(acc[c].get(k, []))):.0f}" for k in ks) + "\n")
print(open(os.path.join(O, "lastaxis_vs_rows_counters.csv")).read())
PYEOF
}

for s in "$@"; do echo "=== $s"; $s; done
