#!/bin/bash
mkdir -p gpurun_out/r04; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04
cd $R
timeout 900 python -m pytest tests/test_accelerate.py tests/test_holder_fast_call.py -m gpu -q > $O/pytest_accel.log 2>&1; tail -15 $O/pytest_accel.log | cut -c1-300
timeout 600 python bench.py --config resnet50 --e2e --steps 100 2>$O/e2e.err | tail -1 > $O/bench_e2e_resnet50.json
timeout 600 python bench.py --config resnet50 --e2e --e2e-lut --steps 100 2>>$O/e2e.err | tail -1 > $O/bench_e2e_resnet50_lut.json
timeout 600 python bench.py --config resnet50 --e2e --batch 32 --steps 50 2>>$O/e2e.err | tail -1 > $O/bench_e2e_resnet50_b32.json
tail -3 $O/e2e.err
python - <<'PY'
import json, os
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r04")
for f in ("bench_e2e_resnet50.json", "bench_e2e_resnet50_lut.json", "bench_e2e_resnet50_b32.json"):
    try:
        d = json.loads(open(os.path.join(O, f)).read())
        print(f, {k: round(v["ms_per_forward"], 3) for k, v in d["modes"].items()}, {k: v.get("quantizer_launches_per_forward") for k, v in d["modes"].items()},
              d["quantized_weights_bit_equal_per_layer_vs_auto_batched"], {k: round(v, 4) for k, v in d["logits_relative_l2_diff_to_per_layer"].items()}, d.get("cpu_baseline", {}).get("gpu_output_bit_equal"), d.get("parity_error"))
    except Exception as e:
        print(f, "ERR", e)
PY
