#!/bin/bash
# round 3: whole GPU suite + the judged line + one line per configuration (after the kernel-argument preload change)
mkdir -p gpurun_out/r03
python -m pytest tests -q -m gpu -x 2>&1 | tail -4 > gpurun_out/r03/pytest_gpu.log; cat gpurun_out/r03/pytest_gpu.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r03/bench_20.json 2> gpurun_out/r03/bench.err
python bench.py > gpurun_out/r03/bench_default.json 2>> gpurun_out/r03/bench.err
: > gpurun_out/r03/bench_all_configs.jsonl
for args in "--config cfg1" "--config cfg2 --batched 16 --steps 100 --warmup 10" "--config cfg3 --batch 1" "--config cfg3 --batch 8" "--config cfg3 --batch 64" "--config cfg3 --batch 64 --stream-depth -1" "--config cfg3 --batch 256" "--config cfg4 --steps 300" "--config cfg5 --steps 300" "--config cfg5 --batched 4 --steps 100 --warmup 10" "--config resnet50 --steps 300 --warmup 20" "--config linear16 --steps 60 --warmup 5"; do
  python bench.py $args --no-cpu >> gpurun_out/r03/bench_all_configs.jsonl 2>> gpurun_out/r03/bench.err
done
python - <<'PY'
import json
for f in ("bench_20.json", "bench_default.json"):
    d = json.load(open("gpurun_out/r03/" + f)); r = d["roofline"]
    print(f, "ms/step %.5f kernel_us %.2f frac %.4f frac_wall %.4f batched16 %.4f bit_equal %s" % (d["ms_per_step"], r["kernel_us"], r["frac"], r["frac_wall"], d["batched_16x4096"]["frac"], d["cpu_baseline"]["gpu_output_bit_equal"]))
for line in open("gpurun_out/r03/bench_all_configs.jsonl"):
    d = json.loads(line); r = d["roofline"]
    print("%-70s us/step %8.3f kernel_us %8.2f frac %.4f frac_wall %.4f  %s" % (d["config"]["workload"][:70] + " " + "x".join(map(str, d["config"]["shape"])), d["ms_per_step"] * 1e3, r["kernel_us"], r["frac"], r["frac_wall"], r["kernel"]))
PY
