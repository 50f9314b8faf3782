#!/bin/bash
# round 4, final evidence at the closing head: e2e lines (incl. auto_captured), full GPU suite in the default mode, the LUT suites with the compact table
mkdir -p gpurun_out/r04; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04
cd $R
python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 600 python bench.py --config resnet50 --e2e --steps 100 2>$O/e2e.err | tail -1 > $O/bench_e2e_resnet50.json
timeout 600 python bench.py --config resnet50 --e2e --e2e-lut --steps 100 2>>$O/e2e.err | tail -1 > $O/bench_e2e_resnet50_lut.json
timeout 600 python bench.py --config resnet50 --e2e --e2e-side 64 --steps 100 2>>$O/e2e.err | tail -1 > $O/bench_e2e_resnet50_64px.json
timeout 600 python bench.py --config resnet50 --e2e --batch 32 --steps 50 2>>$O/e2e.err | tail -1 > $O/bench_e2e_resnet50_b32.json
python - <<'PY'
import json, os
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r04")
for f in ("bench_e2e_resnet50.json", "bench_e2e_resnet50_lut.json", "bench_e2e_resnet50_64px.json", "bench_e2e_resnet50_b32.json"):
    try:
        d = json.loads(open(os.path.join(O, f)).read())
        print(f, {k: round(v["ms_per_forward"], 3) for k, v in d["modes"].items()}, {k: v.get("quantizer_launches_per_forward") for k, v in d["modes"].items()},
              d["quantized_weights_bit_equal_per_layer_vs_auto_batched"], {k: round(v, 4) for k, v in d["logits_relative_l2_diff_to_per_layer"].items()}, d.get("cpu_baseline", {}).get("gpu_output_bit_equal"), d.get("parity_error"))
    except Exception as e:
        print(f, "ERR", e)
PY
timeout 300 python bench.py 2>/dev/null | tail -1 > $O/bench_default.json
timeout 300 python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_20.json
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "all gpu tests rc=$?" >> $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
MCTQ_COMPACT_LUT=1 timeout 1200 python -m pytest tests -m gpu -q -k "lut or Lut or LUT or golden or stream or accelerate" > $O/pytest_gpu_compact_lut.log 2>&1; echo "compact rc=$?" >> $O/pytest_gpu_compact_lut.log; tail -3 $O/pytest_gpu_compact_lut.log
