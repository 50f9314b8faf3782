#!/usr/bin/env python3
"""gpurun_out/lutpmc/group_*.csv (tools/gpu_r04_lut_pmc.sh) -> one table: counter x kernel, average per dispatch.
    python tools/pmc_kernel_table.py gpurun_out/lutpmc profiles/r04/cfg4_lut_vs_affine_counters.csv"""
import csv, glob, os, re, sys
src, dst = sys.argv[1], sys.argv[2]
acc = {}
for path in sorted(glob.glob(os.path.join(src, "group_*.csv"))):
    for r in csv.DictReader(open(path)):
        m = re.search(r"rows_kernel<mctq::(\w+)", r["Kernel_Name"])
        if not m:
            continue
        acc.setdefault(r["Counter_Name"], {}).setdefault(m.group(1), []).append(float(r["Counter_Value"]))
kernels = sorted({k for v in acc.values() for k in v})
os.makedirs(os.path.dirname(dst), exist_ok=True)
with open(dst, "w") as f:
    f.write("counter," + ",".join(f"{k}_avg_per_dispatch" for k in kernels) + ",dispatches\n")
    for c in sorted(acc):
        avgs = [sum(acc[c].get(k, [0.0])) / max(1, len(acc[c].get(k, []))) for k in kernels]
        f.write(c + "," + ",".join(f"{a:.1f}" for a in avgs) + "," + "/".join(str(len(acc[c].get(k, []))) for k in kernels) + "\n")
print(open(dst).read())
