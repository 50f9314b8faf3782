#!/bin/bash
# HISTORICAL (kept as the provenance of profiles/r04/): sections that use --heavy-persistent, MCTQ_COMPACT_LUT, tools/ablate/ or
# the ql_variant experiment codes need the round-4 library (ABI v7, commit f72a681); round 5's driver is tools/gpu_r05.sh.
# How round 4's logs under profiles/r04/ were produced on the MI355X box: `gpurun -- 'bash tools/gpu_r04.sh <section> ...'`
# (several sections per call are fine).  Everything is written under gpurun_out/r04/ and copied to profiles/r04/ by hand.
# rocprofv3 --pmc passes never share a run with other trace domains (tools/gpu_pmc_traffic.sh, tools/gpu_r04_lut_pmc.sh).
mkdir -p gpurun_out/r04; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04
cd $R
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', '| us', round(r['kernel_us'],2), 'frac', round(r['frac'],3), 'wall', round(r['frac_wall'],3), r['kernel'])"; }

evidence() {      # the judged lines + every side line, host overhead, RCCL gather at world size 1, PMC traffic, smoke
  python __graft_entry__.py smoke > $O/smoke.log 2>&1
  timeout 300 python bench.py 2>/dev/null | tail -1 > $O/bench_default.json
  timeout 300 python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_20.json
  : > $O/bench_cfg3.jsonl; : > $O/bench_other_configs.jsonl; : > $O/bench_dtype.jsonl
  for n in 1 8 64; do timeout 300 python bench.py --config cfg3 --batch $n --steps 1000 --warmup 100 2>/dev/null | tail -1 >> $O/bench_cfg3.jsonl; done
  timeout 300 python bench.py --config cfg3 --batch 256 --steps 300 2>/dev/null | tail -1 >> $O/bench_cfg3.jsonl
  timeout 300 python bench.py --config cfg3 --batch 8 --steps 20 --warmup 5 2>/dev/null | tail -1 >> $O/bench_cfg3.jsonl
  for c in cfg4 cfg5 resnet50; do timeout 300 python bench.py --config $c --steps 300 2>/dev/null | tail -1 >> $O/bench_other_configs.jsonl; done
  timeout 300 python bench.py --batched 16 --steps 60 --warmup 5 2>/dev/null | tail -1 >> $O/bench_other_configs.jsonl
  for dt in bf16 f16; do timeout 300 python bench.py --dtype $dt 2>/dev/null | tail -1 >> $O/bench_dtype.jsonl; done
  timeout 300 python bench.py --dtype bf16 --config cfg5 --steps 300 2>/dev/null | tail -1 >> $O/bench_dtype.jsonl
  timeout 300 python bench.py --dtype bf16 --config cfg4 --steps 300 2>/dev/null | tail -1 >> $O/bench_dtype.jsonl
  timeout 300 python bench.py --dtype bf16 --steps 20 --warmup 5 2>/dev/null | tail -1 >> $O/bench_dtype.jsonl
  PORT=$(python -c "import socket; s=socket.socket(); s.bind(('127.0.0.1',0)); print(s.getsockname()[1])")
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $PORT bench.py --gpus 1 --gather --steps 20 --warmup 5 --no-cpu --prewarm-seconds 0.3 --evidence-launches 0 2>/dev/null | tail -1 > $O/bench_torchrun_gather.json
  timeout 300 python tools/host_overhead.py 2>&1 | grep -v amdgpu.ids > $O/host_overhead_per_call.log
  bash tools/gpu_pmc_traffic.sh > $O/pmc_traffic_run.log 2>&1       # then, in the build container: MCTQ_ROUND=r04 python tools/pmc_summarize.py
}

e2e() {           # bench.py --config resnet50 --e2e lines + the host profile and the convolution probe behind their reading
  for v in "resnet50" "resnet50_lut --e2e-lut" "resnet50_64px --e2e-side 64" "resnet50_b32 --batch 32 --steps 50"; do
    set -- $v; name=$1; shift
    timeout 600 python bench.py --config resnet50 --e2e --steps 100 "$@" 2>>$O/e2e.err | tail -1 > $O/bench_e2e_$name.json
  done
  timeout 300 python tools/e2e_host_profile.py 1 2>&1 | grep -v amdgpu.ids > $O/e2e_host_profile_b1.log
  timeout 300 python tools/conv_determinism_probe.py 2>&1 | grep -v amdgpu.ids > $O/conv_determinism_probe.log
}

lut() {           # config 4: counters, ablations (tools/build_variant.py NAME -D... beforehand), table / compact x U x persistent
  bash tools/gpu_r04_lut_pmc.sh                                     # -> gpurun_out/lutpmc; tools/pmc_kernel_table.py
  : > $O/lut_vs_affine_cold.log
  for v in "table MCTQ_COMPACT_LUT=0 0" "compact_U1 MCTQ_COMPACT_LUT=1 1" "compact_U4 MCTQ_COMPACT_LUT=1 4"; do
    set -- $v; echo "== $1" >> $O/lut_vs_affine_cold.log
    env $2 timeout 200 python tools/lut_vs_affine.py 150 $3 2>&1 | grep -v amdgpu.ids >> $O/lut_vs_affine_cold.log
  done
  for v in STAGE1 STAGE4 STAGE1_DIV_LDS; do for hu in 0 1; do
    [ -f $R/tools/ablate/libmctq_hip_$v.so ] || continue
    echo "== ablation $v heavy_unroll=$hu (cold outputs)" >> $O/lut_vs_affine_cold.log
    MCTQ_COMPACT_LUT=0 MCTQ_HIP_LIB=$R/tools/ablate/libmctq_hip_$v.so MCTQ_BINDING=ctypes timeout 200 python tools/lut_vs_affine.py 150 $hu 2>&1 | grep lut >> $O/lut_vs_affine_cold.log
  done; done
  : > $O/cfg4_variants_bench_protocol.log
  run() { local label=$1; shift; local envs=$1; shift
    env $envs timeout 300 python bench.py --config cfg4 --steps 300 --no-cpu --evidence-launches 0 "$@" 2>/dev/null | tail -1 | line "$label" >> $O/cfg4_variants_bench_protocol.log; }
  run "table U=4 (shipped)" MCTQ_COMPACT_LUT=0
  run "compact U=4" MCTQ_COMPACT_LUT=1
  for u in 1 2; do run "compact U=$u" MCTQ_COMPACT_LUT=1 --heavy-unroll $u; done
  for u in 1 2 4; do run "compact persistent U=$u" MCTQ_COMPACT_LUT=1 --heavy-persistent 1 --heavy-unroll $u; done
  run "table persistent U=4" MCTQ_COMPACT_LUT=0 --heavy-persistent 1 --heavy-unroll 4
}

shapes() {        # launch-shape probes: rowsteps_kernel vs rows_kernel (sustained), 16-bit A/B under bench.py, flat-kernel grid rounds
  timeout 900 python tools/rowsteps_probe.py 2>&1 | grep -v amdgpu.ids > $O/rowsteps_probe_sustained.log
  : > $O/bf16_tuning.log
  for rep in 1 2; do for rs in 0 1 2; do
    timeout 300 python bench.py --no-cpu --evidence-launches 0 --dtype bf16 --rowsteps $rs 2>/dev/null | tail -1 | line "bf16 4096^2 rowsteps=$rs" >> $O/bf16_tuning.log
  done; done
  : > $O/flat_grid_rounds.log
  for n in 40 48 52 55 56 57 60 64 72 80 96 111 112 113 128; do for u in 4 8 2; do
    timeout 200 python bench.py --config cfg3 --batch $n --stream-depth -1 --steps 400 --warmup 50 --prewarm-seconds 0.5 --no-cpu --no-eager-extra --evidence-launches 0 --unroll $u 2>/dev/null | tail -1 | line "N $n unroll $u" >> $O/flat_grid_rounds.log
  done; done
}

suites() {        # the GPU suite in every mode + the seeded fuzz suites on further seeds
  timeout 1800 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log
  MCTQ_COMPACT_LUT=1 timeout 1200 python -m pytest tests -m gpu -q -k "lut or Lut or LUT or golden or stream or accelerate" > $O/pytest_gpu_compact_lut.log 2>&1; echo "rc=$?" >> $O/pytest_gpu_compact_lut.log
  MCTQ_BINDING=ctypes timeout 1800 python -m pytest tests -m gpu -q > $O/pytest_gpu_ctypes.log 2>&1; echo "rc=$?" >> $O/pytest_gpu_ctypes.log
  MCTQ_ROCTX=1 timeout 1800 python -m pytest tests -m gpu -q -k "not every_float and not 2_32" > $O/pytest_gpu_roctx.log 2>&1; echo "rc=$?" >> $O/pytest_gpu_roctx.log
  SEEDS="21 22 23" bash tools/gpu_fuzz_soak.sh
  MCTQ_COMPACT_LUT=1 SEEDS="24 25" bash tools/gpu_fuzz_soak.sh | sed 's/^/compact: /' > $O/fuzz_soak_compact.log
  tail -2 $O/pytest_gpu.log $O/pytest_gpu_compact_lut.log $O/pytest_gpu_ctypes.log $O/pytest_gpu_roctx.log; cat $O/fuzz_soak.log $O/fuzz_soak_compact.log
}

for section in "$@"; do $section; done
