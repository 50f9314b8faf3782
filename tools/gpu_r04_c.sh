#!/bin/bash
# round 4, third GPU trip: compact LUT table (tests, timing, counters), conv determinism probe, cfg4 bench, full suite
mkdir -p gpurun_out/r04c; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04c
cd $R
timeout 1500 python -m pytest tests/test_gpu_lut_compact.py tests/test_accelerate.py tests/test_holder_fast_call.py tests/test_gpu_affine_rowsteps.py -m gpu -q -x > $O/pytest_new.log 2>&1; echo "new tests rc=$?" >> $O/pytest_new.log
tail -12 $O/pytest_new.log
timeout 300 python tools/host_overhead.py 2>&1 | grep -v amdgpu.ids > $O/host_overhead.log; grep -E "Holder|AffinePlan|ActivationUniform|LutPOT" $O/host_overhead.log
for hu in 0 1 2 4; do echo "== compact, heavy_unroll=$hu" >> $O/lut_compact_timing.log; timeout 200 python tools/lut_vs_affine.py 150 $hu 2>&1 | grep -v amdgpu.ids >> $O/lut_compact_timing.log; done
echo "== full table (MCTQ_COMPACT_LUT=0)" >> $O/lut_compact_timing.log; MCTQ_COMPACT_LUT=0 timeout 200 python tools/lut_vs_affine.py 150 0 2>&1 | grep -v amdgpu.ids >> $O/lut_compact_timing.log
cat $O/lut_compact_timing.log
timeout 300 python tools/conv_determinism_probe.py 2>&1 | grep -v amdgpu.ids > $O/conv_determinism_probe.log; cat $O/conv_determinism_probe.log
timeout 300 python bench.py --config cfg4 --steps 300 2>/dev/null | tail -1 > $O/bench_cfg4.json
timeout 300 python bench.py --config cfg4 --dtype bf16 --steps 300 2>/dev/null | tail -1 > $O/bench_cfg4_bf16.json
python - <<'PY'
import json, os
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r04c")
for f in ("bench_cfg4.json", "bench_cfg4_bf16.json"):
    d = json.loads(open(os.path.join(O, f)).read()); r = d["roofline"]
    print(f, "us", round(r["kernel_us"], 2), "frac", round(r["frac"], 3), r["kernel"], "parity", d.get("cpu_baseline", {}).get("gpu_output_bit_equal"))
PY
rm -rf gpurun_out/lutpmc; bash tools/gpu_r04_lut_pmc.sh
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "all gpu tests rc=$?" >> $O/pytest_gpu.log; tail -8 $O/pytest_gpu.log
