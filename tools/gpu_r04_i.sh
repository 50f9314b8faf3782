#!/bin/bash
mkdir -p gpurun_out/r04i; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04i
cd $R
timeout 600 python -m pytest tests/test_gpu_affine_rowsteps.py tests/test_gpu_multigpu.py tests/test_holder_fast_call.py -m gpu -q > $O/pytest_sel.log 2>&1; tail -3 $O/pytest_sel.log
for dt in bf16 f16; do timeout 300 python bench.py --dtype $dt 2>/dev/null | tail -1 >> $O/bench_dtype.jsonl; done
timeout 300 python bench.py --dtype bf16 --config cfg5 --steps 300 2>/dev/null | tail -1 >> $O/bench_dtype.jsonl
timeout 300 python bench.py --dtype bf16 --config cfg4 --steps 300 2>/dev/null | tail -1 >> $O/bench_dtype.jsonl
timeout 300 python bench.py --dtype bf16 --steps 20 --warmup 5 2>/dev/null | tail -1 >> $O/bench_dtype.jsonl
timeout 300 python tools/host_overhead.py 2>&1 | grep -v amdgpu.ids > $O/host_overhead_per_call.log; grep -E "Holder|AffinePlan|quantizer\(x\)" $O/host_overhead_per_call.log
MCTQ_PMC_ONLY="cfg2_bf16 cfg2_f16 cfg5_bf16 cfg3_n64 cfg3_n8" bash tools/gpu_pmc_traffic.sh > $O/pmc_run.log 2>&1; tail -12 $O/pmc_run.log | cut -c1-250
python - <<'PY'
import json, os
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r04i")
for ln in open(os.path.join(O, "bench_dtype.jsonl")):
    if ln.startswith("{"):
        d = json.loads(ln); r = d["roofline"]
        print(d["dtype"], d["steps"], d["config"]["workload"][:44], "| us", round(r["kernel_us"], 2), "frac", round(r["frac"], 3), "wall", round(r["frac_wall"], 3), r["kernel"][:60], "| parity", d.get("cpu_baseline", {}).get("gpu_output_bit_equal"), r.get("traffic_source"))
PY
