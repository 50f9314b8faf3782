#!/bin/bash
# round 4, second GPU trip: new tests, host overhead, staging ablations of the LUT table kernel, rowsteps probe, dtype / cfg3 / e2e bench lines
mkdir -p gpurun_out/r04b; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04b
cd $R
timeout 900 python -m pytest tests/test_accelerate.py tests/test_holder_fast_call.py tests/test_gpu_affine_rowsteps.py -m gpu -q > $O/pytest_new.log 2>&1; echo "new tests rc=$?" >> $O/pytest_new.log
tail -30 $O/pytest_new.log
timeout 300 python tools/host_overhead.py > $O/host_overhead.log 2>&1; cat $O/host_overhead.log
for v in "" STAGE1 STAGE2 STAGE3 STAGE4 STAGE1_DIV STAGE1_DIV_LDS; do
  L=""; [ -n "$v" ] && L=$R/tools/ablate/libmctq_hip_$v.so
  for hu in 0 2 1; do
    [ -z "$v" ] && [ $hu != 0 ] && continue
    echo "== variant ${v:-shipped} heavy_unroll=$hu" >> $O/lut_staging_ablation.log
    if [ -n "$L" ]; then MCTQ_HIP_LIB=$L MCTQ_BINDING=ctypes timeout 200 python tools/lut_vs_affine.py 150 $hu 2>&1 | grep -v amdgpu.ids >> $O/lut_staging_ablation.log
    else MCTQ_BINDING=ctypes timeout 200 python tools/lut_vs_affine.py 150 $hu 2>&1 | grep -v amdgpu.ids >> $O/lut_staging_ablation.log; fi
  done
done
cat $O/lut_staging_ablation.log
timeout 600 python tools/rowsteps_probe.py 2>&1 | grep -v amdgpu.ids > $O/rowsteps_probe.log; cat $O/rowsteps_probe.log
for dt in bf16 f16; do timeout 300 python bench.py --dtype $dt --steps 1000 --warmup 100 2>/dev/null | tail -1 >> $O/bench_dtype.jsonl; done
timeout 300 python bench.py --dtype bf16 --config cfg5 --steps 300 2>/dev/null | tail -1 >> $O/bench_dtype.jsonl
timeout 300 python bench.py --dtype bf16 --config cfg4 --steps 300 2>/dev/null | tail -1 >> $O/bench_dtype.jsonl
for n in 1 8 64; do timeout 300 python bench.py --config cfg3 --batch $n --steps 1000 --warmup 100 2>/dev/null | tail -1 >> $O/bench_cfg3.jsonl; done
timeout 600 python bench.py --config resnet50 --e2e --steps 100 2>$O/e2e.err | tail -1 > $O/bench_e2e_resnet50.json
timeout 600 python bench.py --config resnet50 --e2e --e2e-lut --steps 100 2>>$O/e2e.err | tail -1 > $O/bench_e2e_resnet50_lut.json
timeout 600 python bench.py --config resnet50 --e2e --batch 32 --steps 50 2>>$O/e2e.err | tail -1 > $O/bench_e2e_resnet50_b32.json
tail -5 $O/e2e.err
python - <<'PY'
import json, os
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r04b")
for f in ("bench_dtype.jsonl", "bench_cfg3.jsonl"):
    for ln in open(os.path.join(O, f)):
        if ln.startswith("{"):
            d = json.loads(ln); r = d["roofline"]
            print(f, d["dtype"], d["config"]["workload"][:40], "us", round(r["kernel_us"], 2), "frac", round(r["frac"], 3), "wall", round(r["frac_wall"], 3), r["kernel"],
                  "eager", d.get("eager_us_per_batch"), d.get("eager_frac"), d.get("eager_host_us_per_call"), "parity", d.get("cpu_baseline", {}).get("gpu_output_bit_equal"))
for f in ("bench_e2e_resnet50.json", "bench_e2e_resnet50_lut.json", "bench_e2e_resnet50_b32.json"):
    try:
        d = json.loads(open(os.path.join(O, f)).read())
        print(f, {k: round(v["ms_per_forward"], 3) for k, v in d["modes"].items()}, {k: v.get("quantizer_launches_per_forward") for k, v in d["modes"].items()},
              d["outputs_bit_equal_to_per_layer"], "roofline", round(d["roofline"]["kernel_us"], 1), round(d["roofline"]["frac"], 3), d.get("cpu_baseline", {}).get("gpu_output_bit_equal"), d.get("parity_error"))
    except Exception as e:
        print(f, "ERR", e)
PY
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "all gpu tests rc=$?" >> $O/pytest_gpu.log; tail -8 $O/pytest_gpu.log
