#!/bin/bash
# round 4, trip f: full GPU suite, e2e host profile + e2e bench lines, conv probe, PMC traffic for the 16-bit lines
mkdir -p gpurun_out/r04f; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04f
cd $R
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "all gpu tests rc=$?" >> $O/pytest_gpu.log; tail -8 $O/pytest_gpu.log
timeout 300 python tools/e2e_host_profile.py 1 2>&1 | grep -v amdgpu.ids > $O/e2e_host_profile_b1.log; head -60 $O/e2e_host_profile_b1.log
timeout 300 python tools/conv_determinism_probe.py 2>&1 | grep -v amdgpu.ids > $O/conv_determinism_probe.log
timeout 600 python bench.py --config resnet50 --e2e --steps 100 2>$O/e2e.err | tail -1 > $O/bench_e2e_resnet50.json
timeout 600 python bench.py --config resnet50 --e2e --e2e-lut --steps 100 2>>$O/e2e.err | tail -1 > $O/bench_e2e_resnet50_lut.json
timeout 600 python bench.py --config resnet50 --e2e --e2e-side 64 --steps 100 2>>$O/e2e.err | tail -1 > $O/bench_e2e_resnet50_64px.json
python - <<'PY'
import json, os
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r04f")
for f in ("bench_e2e_resnet50.json", "bench_e2e_resnet50_lut.json", "bench_e2e_resnet50_64px.json"):
    try:
        d = json.loads(open(os.path.join(O, f)).read())
        print(f, {k: round(v["ms_per_forward"], 3) for k, v in d["modes"].items()}, {k: v.get("quantizer_launches_per_forward") for k, v in d["modes"].items()},
              d["quantized_weights_bit_equal_per_layer_vs_auto_batched"], d["logits_max_abs_diff_to_per_layer"], "roofline", round(d["roofline"]["kernel_us"], 1), round(d["roofline"]["frac"], 3), d.get("cpu_baseline", {}).get("gpu_output_bit_equal"), d.get("parity_error"))
    except Exception as e:
        print(f, "ERR", e)
PY
