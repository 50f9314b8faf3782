#!/usr/bin/env python3
"""Summarise gpurun_out/pmc/<config>/ (written by tools/gpu_pmc_traffic.sh on the GPU box) into
profiles/pmc_traffic.json and profiles/r03/: per config the kernel VARIANT (mctq_last_launch, taken from the JSON line
of the same profiled bench run), rocprof's kernel symbol, the average FETCH_SIZE / WRITE_SIZE per dispatch, the gfx950
correction (FETCH_SIZE x 2 for wide coalesced reads, guide MI355X_MICROARCH.md "HBM"), and the git head they were
taken at and the library build (mctq_build_id) they were taken on.  bench.py emits roofline.traffic only when its own variant
string AND its library's build id equal the recorded ones."""
import csv, json, os, shutil, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(REPO, "gpurun_out", "pmc")
DST = os.path.join(REPO, "profiles", os.environ.get("MCTQ_ROUND", "r06"))
os.makedirs(DST, exist_ok=True)
head = subprocess.run(["git", "rev-parse", "--short=12", "HEAD"], cwd=REPO, capture_output=True, text=True).stdout.strip()

def bench_line(path):
    for line in reversed(open(path).read().splitlines()):
        if line.startswith("{"):
            return json.loads(line)
    return None

def counter_avg(path, name):
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == name]
    by_kernel = {}
    for r in rows:
        by_kernel.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
    kern = max(by_kernel, key=lambda k: len(by_kernel[k]))        # the judged kernel dominates the dispatch count
    vals = by_kernel[kern]
    return kern, sum(vals) / len(vals), len(vals)

out = {}
for cfg in sorted(os.listdir(SRC)):
    d = os.path.join(SRC, cfg)
    if not os.path.isdir(d) or not os.path.exists(os.path.join(d, "FETCH_SIZE.csv")):
        continue
    if sum(1 for _ in open(os.path.join(d, "FETCH_SIZE.csv"))) < 2:
        print(cfg, "no counter rows (skipped)")
        continue
    line = bench_line(os.path.join(d, "bench_FETCH_SIZE.log")) or {}
    line_w = bench_line(os.path.join(d, "bench_WRITE_SIZE.log")) or {}
    variant = line.get("roofline", {}).get("kernel")
    assert variant and variant == line_w.get("roofline", {}).get("kernel"), (cfg, variant)
    kern, fetch_kb, nf = counter_avg(os.path.join(d, "FETCH_SIZE.csv"), "FETCH_SIZE")
    kern_w, write_kb, nw = counter_avg(os.path.join(d, "WRITE_SIZE.csv"), "WRITE_SIZE")
    assert kern == kern_w, (kern, kern_w)
    alg = line["roofline"]["algorithmic_bytes_per_launch"]
    hbm = (2.0 * fetch_kb + write_kb) * 1024.0
    rows = list(csv.DictReader(open(os.path.join(d, "kernel_stats.csv"))))
    stats = next((r for r in rows if r["Name"] == kern), rows[0])      # the judged kernel's row of the stats pass
    key = line.get("roofline", {}).get("traffic_key", cfg)       # the key bench.py will look its own run up under
    bid = line.get("config", {}).get("library_build_id")
    assert bid and bid == line_w.get("config", {}).get("library_build_id"), (cfg, bid)
    out[key] = {"variant": variant, "kernel": kern, "git_head": head, "build_id": bid, "profiled_as": cfg,
                "FETCH_SIZE_KB_raw": fetch_kb, "WRITE_SIZE_KB": write_kb,
                "correction": "gfx950: FETCH_SIZE reports 1/2 of wide (16 B/lane) streaming reads -> doubled (guide MI355X_MICROARCH.md, HBM section); WRITE_SIZE exact",
                "hbm_bytes_per_launch": hbm, "algorithmic_bytes_per_launch": alg, "traffic_over_algorithmic": hbm / alg,
                "rocprof_kernel_avg_ns": float(stats["AverageNs"]), "rocprof_kernel_calls": int(stats["Calls"]),
                "passes": f"separate rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE runs of the judged bench command "
                          f"(tools/gpu_pmc_traffic.sh), {nf}/{nw} dispatches averaged"}
    for f in ("kernel_stats.csv", "FETCH_SIZE.csv", "WRITE_SIZE.csv"):
        shutil.copy(os.path.join(d, f), os.path.join(DST, f"{cfg}_{f}"))
    print(cfg, variant, f"traffic/alg = {hbm / alg:.4f}", f"rocprof avg {float(stats['AverageNs']) / 1e3:.2f} us over {stats['Calls']} calls")
with open(os.path.join(REPO, "profiles", "pmc_traffic.json"), "w") as f:
    json.dump(out, f, indent=1)

# stall counters of the headline kernel and of the batched launch -> one table each
for cfg, needle, fname in (("cfg2", "rows_kernel", "cfg2_rows_kernel_stall_counters.csv"),
                           ("cfg2_batched16", "batched_table_kernel", "cfg2_batched16_table_kernel_stall_counters.csv"),
                           ("cfg2_bf16", "shortrows_kernel", "cfg2_bf16_shortrows_kernel_stall_counters.csv"),
                           ("sym_4096x4096_axis1_bf16", "lastaxis_kernel", "sym_4096x4096_axis1_bf16_lastaxis_kernel_stall_counters.csv"),
                           ("sym_65536x200_axis1_bf16", "lastaxis_kernel", "sym_65536x200_axis1_bf16_lastaxis_kernel_stall_counters.csv"),
                           ("sym_16384x1020_axis0_bf16", "shortrows_kernel", "sym_16384x1020_axis0_bf16_shortrows_kernel_stall_counters.csv"),
                           ("sym_1048576x16_axis0_bf16", "shortrows_kernel", "sym_1048576x16_axis0_bf16_shortrows_kernel_stall_counters.csv")):
    stall = {}
    d = os.path.join(SRC, cfg)
    if not os.path.isdir(d):
        continue
    for fn in sorted(os.listdir(d)):
        if fn.startswith("stall_") and fn.endswith(".csv"):
            for r in csv.DictReader(open(os.path.join(d, fn))):
                if needle in r["Kernel_Name"]:
                    stall.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    if stall:
        with open(os.path.join(DST, fname), "w") as f:
            f.write("counter,avg_per_dispatch,dispatches\n")
            for k in sorted(stall):
                f.write(f"{k},{sum(stall[k]) / len(stall[k]):.1f},{len(stall[k])}\n")
        print(fname)
        print(open(os.path.join(DST, fname)).read())
if os.path.exists(os.path.join(SRC, "available_counters.txt")):
    shutil.copy(os.path.join(SRC, "available_counters.txt"), os.path.join(DST, "rocprofv3_available_counters.txt"))
