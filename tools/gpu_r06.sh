#!/bin/bash
# How round 6's logs under profiles/r06/ were produced on the MI355X box: `gpurun -- 'bash tools/gpu_r06.sh <section> ...'`.
# Everything is written under gpurun_out/r06/ and copied to profiles/r06/ by hand.  (Counter passes: tools/gpu_pmc_traffic.sh, then
# MCTQ_ROUND=r06 python tools/pmc_summarize.py in the build container.)
mkdir -p gpurun_out/r06; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06
cd $R
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', '| us', round(r['kernel_us'],2), 'frac', round(r['frac'],3), 'wall', round(r['frac_wall'],3), r['kernel'], '| traffic', r.get('traffic'), '| parity', d.get('ranks_parity_ok'), d.get('cpu_baseline', {}).get('gpu_output_bit_equal'))"; }

suite() {         # the GPU suite at the current head
  timeout 1800 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log
  tail -3 $O/pytest_gpu.log
}

bench() {         # judged line + side lines (the default line now also carries rank_devices and the sharded config-5 leg at N = 1)
  python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -2 $O/smoke.log
  timeout 400 python bench.py 2>$O/bench_default.err | tail -1 > $O/bench_default.json
  timeout 400 python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_20.json
  : > $O/bench_other_configs.jsonl; : > $O/bench_dtype.jsonl; : > $O/bench_cfg3.jsonl; : > $O/bench_shapes.jsonl
  for c in cfg4 cfg5 resnet50; do timeout 400 python bench.py --config $c --steps 300 --no-sharded-extra 2>/dev/null | tail -1 >> $O/bench_other_configs.jsonl; done
  for dt in bf16 f16; do timeout 400 python bench.py --dtype $dt --no-sharded-extra 2>/dev/null | tail -1 >> $O/bench_dtype.jsonl; done
  timeout 400 python bench.py --dtype bf16 --config cfg5 --steps 300 --no-sharded-extra 2>/dev/null | tail -1 >> $O/bench_dtype.jsonl
  timeout 400 python bench.py --dtype bf16 --config cfg4 --steps 300 --no-sharded-extra 2>/dev/null | tail -1 >> $O/bench_dtype.jsonl
  for n in 1 8 64; do timeout 400 python bench.py --config cfg3 --batch $n --steps 1000 --warmup 100 --no-sharded-extra 2>/dev/null | tail -1 >> $O/bench_cfg3.jsonl; done
  # per-tensor launches of one round (flat_paced_kernel) beside the same launch with the key off
  : > $O/bench_paced.jsonl
  for pv in 1 0; do
    timeout 400 python bench.py --config cfg3 --batch 50 --stream-depth -1 --steps 1000 --warmup 100 --paced $pv --no-sharded-extra 2>/dev/null | tail -1 >> $O/bench_paced.jsonl
    timeout 400 python bench.py --config cfg3 --batch 100 --dtype bf16 --stream-depth -1 --steps 1000 --warmup 100 --paced $pv --no-sharded-extra 2>/dev/null | tail -1 >> $O/bench_paced.jsonl
  done
  # the launch shapes VERDICT r05 #1 names, under the judged protocol (cold ring, 1 s pre-warm, events inside the timed region)
  for sa in "4096x4096 1" "65536x200 1" "16384x1020 0" "1048576x16 0" "16384x1024 0" "65536x256 0" "64x56x56x256 3" "4096x4100 0"; do
    set -- $sa
    timeout 400 python bench.py --config sym --shape $1 --axis $2 --dtype bf16 --no-sharded-extra 2>/dev/null | tail -1 >> $O/bench_shapes.jsonl
  done
  timeout 400 python bench.py --config sym --shape 4096x4096 --axis 1 --dtype bf16 --cached-store-max-mb 0 --no-sharded-extra 2>/dev/null | tail -1 >> $O/bench_shapes.jsonl
  timeout 400 python bench.py --config sym --shape 4096x4096 --axis 1 --no-sharded-extra 2>/dev/null | tail -1 >> $O/bench_shapes.jsonl
  timeout 400 python bench.py --config sym --shape 1048576x16 --axis 0 --no-sharded-extra 2>/dev/null | tail -1 >> $O/bench_shapes.jsonl
  for f in bench_default.json bench_20.json; do line $f < $O/$f; done
  for f in bench_other_configs bench_dtype bench_cfg3 bench_paced bench_shapes; do while read -r l; do echo "$l" | line $f; done < $O/$f.jsonl; done
}

rehearse() {      # the N > 1 entry path on the one GPU: must refuse without --allow-gloo (two ranks, one device), and say so with it
  ( time MCTQ_BENCH_WRAP_DEVICES=1 timeout 600 python bench.py --gpus 2 --steps 20 --warmup 5 ) > $O/bench_gpus2_wrapped_no_allow.log 2>&1
  echo "rc=$? (expected non-zero)" >> $O/bench_gpus2_wrapped_no_allow.log
  ( time MCTQ_BENCH_WRAP_DEVICES=1 timeout 1200 python bench.py --gpus 2 --steps 20 --warmup 5 --allow-gloo ) > $O/bench_gpus2_wrapped_allow.log 2>&1
  echo "rc=$?" >> $O/bench_gpus2_wrapped_allow.log
  tail -4 $O/bench_gpus2_wrapped_no_allow.log | cut -c1-400
  python -c "
import json
for l in open('$O/bench_gpus2_wrapped_allow.log'):
    if l.startswith('{'):
        d=json.loads(l); print({k: d.get(k) for k in ('n_gpus','ranks_seen','rank_devices','devices_distinct','ranks_parity_ok','sharded_cfg5_compute_elems_per_s','sharded_cfg5_compute_plus_allgather_elems_per_s','sharded_cfg5_allgather_gbs_per_link')}, d['config']['control_plane'])"
}

soak() {          # closing soak at the final head: the GPU suite twice more, the judged command three times
  : > $O/closing_soak.log
  for i in 1 2; do timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -1 | sed "s/^/suite run $i: /" >> $O/closing_soak.log; done
  for i in 1 2 3; do timeout 400 python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 | line "bench --steps 20 run $i" >> $O/closing_soak.log; done
  cat $O/closing_soak.log
}

sweep() {         # tools/sweep_shapes.py with the round-6 protocol (the three storage types)
  timeout 900 python tools/sweep_shapes.py --dtypes f32,bf16,f16 > $O/sweep_shapes.log 2>&1; grep -c . $O/sweep_shapes.log; grep "spread >" $O/sweep_shapes.log | head
}

modes() {         # the GPU suite through the other binding / with roctx ranges, and the seeded fuzz suites on further seeds
  MCTQ_BINDING=ctypes timeout 1800 python -m pytest tests -m gpu -q > $O/pytest_gpu_ctypes.log 2>&1; echo "rc=$?" >> $O/pytest_gpu_ctypes.log; tail -2 $O/pytest_gpu_ctypes.log
  MCTQ_ROCTX=1 timeout 1800 python -m pytest tests -m gpu -q -k "not every_float and not 2_32" > $O/pytest_gpu_roctx.log 2>&1; echo "rc=$?" >> $O/pytest_gpu_roctx.log; tail -2 $O/pytest_gpu_roctx.log
  MCTQ_ROUND=r06 SEEDS="${SEEDS:-55 56 57 58}" bash tools/gpu_fuzz_soak.sh
}

e2e() {           # bench.py --config resnet50 --e2e at the final build
  timeout 600 python bench.py --config resnet50 --e2e --steps 100 2>$O/e2e.err | tail -1 > $O/bench_e2e_resnet50.json
  python -c "
import json; d=json.load(open('$O/bench_e2e_resnet50.json')); print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in d.items() if k in ('value', 'ms_per_step') or k.startswith('ms_')})" 2>/dev/null || head -c 600 $O/bench_e2e_resnet50.json
}

pacedlines() {    # per-tensor launches at the sizes the window was found on, under the judged protocol, key on / off
  : > $O/bench_paced_pertensor.jsonl
  for sd in "4096x4096 bf16" "4096x4096 f16" "3584x4096 bf16" "2048x4096 f32" "1792x4096 f32"; do
    set -- $sd
    for pv in 1 0; do
      timeout 400 python bench.py --config sym --shape $1 --per-tensor --dtype $2 --paced $pv --no-sharded-extra 2>/dev/null | tail -1 >> $O/bench_paced_pertensor.jsonl
    done
  done
  while read -r l; do echo "$l" | line bench_paced_pertensor; done < $O/bench_paced_pertensor.jsonl
  # ... and symmetric float32 per-channel launches of the window (shortrows_kernel with paced stores against the old routes)
  : > $O/bench_paced_rows.jsonl
  for sh in 2048x4096 8192x1024 32768x256; do
    for pv in 1 0; do
      timeout 400 python bench.py --config sym --shape $sh --axis 0 --paced $pv --no-sharded-extra 2>/dev/null | tail -1 >> $O/bench_paced_rows.jsonl
    done
  done
  while read -r l; do echo "$l" | line bench_paced_rows; done < $O/bench_paced_rows.jsonl
}

abrows() {        # per-channel launches of the window through the shipped library, key off against on
  timeout 500 python tools/ab_probe.py --a shipped:paced=0 --b shipped --cases pacedrows32,pacedrows16 > $O/ab_paced_rows.log 2>&1; grep -c "B/A" $O/ab_paced_rows.log
}

abpaced() {       # the per-tensor window through the shipped library: paced 0 against 2 (every size through flat_paced_kernel)
  timeout 800 python tools/ab_probe.py --a shipped:paced=0 --b shipped:paced=2 --cases paced16,paced32 > $O/ab_paced.log 2>&1; grep -c "B/A" $O/ab_paced.log
}

for s in "$@"; do $s; done
