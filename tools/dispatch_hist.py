#!/usr/bin/env python3
"""Per-dispatch durations of the judged launch, joined with what the host knows about each launch.

VERDICT r04 item 2: rocprofv3's kernel statistics of the headline run show MinNs 20 160 against an average of 21 994 --
is the fast population reproducible and controllable (ring slot, output placement, the gap in front of the launch), or
is it the warm-up / clock / cache residue of a few dispatches?

Two halves, one file:

  run      (under ``rocprofv3 --kernel-trace -- python3 tools/dispatch_hist.py run ...``): issues the launches of
           bench.py's step loop phase by phase and writes one host record per launch (phase, index, ring slot, input and
           output address) to ``--log``.  Phases:
             bench     bench.py's own step: ``ys[slot] = quantizer(xs[slot])`` over the cold ring, back to back
             fixed     the same launches into a ring of pre-allocated outputs (C ABI, no allocator in the loop)
             ring9     as fixed, ring of 9 pairs
             warm      one buffer pair (working set inside the Infinity Cache)
             synced    a device synchronize after every launch (no neighbour in flight: the kernel alone)
             gap5/20   a host-side spin of 5 / 20 us between launches (stream idle in front of every launch)
  analyze  joins the k-th host record with the k-th dispatch of the judged kernel in the trace (both are in issue order
           on one stream) and writes the per-dispatch table + a histogram per phase / slot / output address.

Works for ``--config cfg2|cfg4|cfg5`` and ``--dtype f32|bf16|f16``; the fixed / ring9 / warm / synced / gap phases use the
public class as well (the outputs then come from torch's allocator: their addresses are recorded all the same) unless the
configuration is the affine cfg2 / cfg5, where the C ABI is called with explicit output pointers.
"""
import argparse
import csv
import glob
import gzip
import json
import os
import statistics
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def run(args):
    import ctypes

    import torch

    import mct_quantizers_amd as mq
    from mct_quantizers_amd import workloads
    from mct_quantizers_amd.hip import native

    lib = native.load()
    dev = torch.device("cuda", 0)
    x_np = workloads.make_input(args.config)
    wl = workloads.make_workload(args.config, x_np)
    quantizer = getattr(mq.pytorch_quantizers, wl.quantizer)(**wl.kwargs)
    tdtype = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}[args.dtype]
    x0 = torch.from_numpy(x_np).to(dev).to(tdtype)
    out_dtype = torch.float32 if "LUT" in wl.quantizer else tdtype
    nbytes = x0.numel() * (x0.element_size() + torch.empty(0, dtype=out_dtype).element_size())
    ring = max(2, -(-(512 << 20) // nbytes) + 1)
    recs = []
    affine_abi = args.config in ("cfg2", "cfg5")

    def pairs(n):
        return [x0] + [x0.clone() for _ in range(n - 1)], [torch.empty(x0.shape, dtype=out_dtype, device=dev) for _ in range(n)]

    if affine_abi:
        q0 = quantizer(x0)                                     # moves the parameter tensors to the device
        recs.append(("setup", 0, 0, x0.data_ptr(), q0.data_ptr()))
        del q0
        scales = quantizer.scales.flatten().contiguous()
        assert scales.device.type == "cuda"
        rows, cols = x0.shape
        dt_code = {"f32": 0, "f16": 1, "bf16": 2}[args.dtype]
        fn = lib.mctq_fq_per_channel
        qmin, qmax = int(quantizer.min_quantized_domain), int(quantizer.max_quantized_domain)
        stream = torch.cuda.current_stream().cuda_stream

        def abi_call(x, y):
            rc = fn(x.data_ptr(), y.data_ptr(), 1, rows, cols, dt_code, scales.data_ptr(), None, qmin, qmax, stream)
            native.check(rc, "mctq_fq_per_channel")
    else:
        def abi_call(x, y):                                    # public class; the allocator's output address is recorded
            return quantizer(x)

    def phase(name, n_launch, n_ring, fixed, after=None):
        xs, ys = pairs(n_ring)
        if not fixed:
            ys = [None] * n_ring
        torch.cuda.synchronize()
        for i in range(n_launch):
            s = i % n_ring
            if fixed and affine_abi:
                abi_call(xs[s], ys[s])
                yp = ys[s].data_ptr()
            else:
                ys[s] = quantizer(xs[s])
                yp = ys[s].data_ptr()
            recs.append((name, i, s, xs[s].data_ptr(), yp))
            if after is not None:
                after()
            elif i % 256 == 255:
                torch.cuda.synchronize()                       # bounded queue (as bench.py's pre-warm)
        torch.cuda.synchronize()
        del xs, ys
        torch.cuda.empty_cache()

    def spin(us):
        def f():
            t = time.perf_counter() + us * 1e-6
            while time.perf_counter() < t:
                pass
        return f

    # clocks up first (bench.py's declared pre-warm is 1 s; the records of this phase are kept: "time since start")
    n = args.launches
    phase("bench", 4 * n, ring, fixed=False)
    phase("fixed", n, ring, fixed=True)
    phase("ring9", n, 9, fixed=True)
    phase("warm", n, 1, fixed=True)
    phase("synced", min(n, 1000), ring, fixed=True, after=torch.cuda.synchronize)
    phase("gap5", min(n, 1000), ring, fixed=True, after=spin(5 + nbytes / 6e6))     # kernel time + 5 us
    phase("gap20", min(n, 1000), ring, fixed=True, after=spin(20 + nbytes / 6e6))
    phase("bench2", n, ring, fixed=False)
    with open(args.log, "w") as f:
        json.dump({"config": args.config, "dtype": args.dtype, "kernel": native.last_launch(), "ring": ring,
                   "bytes_per_launch": nbytes, "build_id": lib.mctq_build_id().decode(),
                   "records": recs}, f)
    print(f"{len(recs)} launches recorded; last variant {native.last_launch()}")


def find_trace(d):
    hits = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))
    if not hits:
        raise SystemExit(f"no *kernel_trace.csv under {d}")
    return hits[-1]


def analyze(args):
    with open(args.log) as f:
        log = json.load(f)
    recs = log["records"]
    shape_name = log["kernel"].split("<")[0]                   # rows_kernel / rowsteps_kernel / ...
    rows = []
    with open(find_trace(args.trace)) as f:
        for r in csv.DictReader(f):
            if shape_name in r["Kernel_Name"] and "batched" not in r["Kernel_Name"]:
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    rows.sort()
    if len(rows) != len(recs):
        print(f"warning: {len(rows)} dispatches of {shape_name} in the trace, {len(recs)} host records; joining the first "
              f"{min(len(rows), len(recs))} in order", file=sys.stderr)
    n = min(len(rows), len(recs))
    t0 = rows[0][0]
    table = []
    for k in range(n):
        ph, i, slot, xp, yp = recs[k]
        st, en = rows[k]
        gap = st - rows[k - 1][1] if k else 0
        table.append((k, ph, i, slot, xp, yp, (st - t0) / 1e3, en - st, gap))
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    with gzip.open(args.out + "_per_dispatch.csv.gz", "wt", newline="") as f:
        w = csv.writer(f)
        w.writerow(["k", "phase", "i", "slot", "x_addr", "y_addr", "start_us", "dur_ns", "gap_before_ns"])
        for t in table:
            w.writerow([t[0], t[1], t[2], t[3], hex(t[4]), hex(t[5]), f"{t[6]:.1f}", t[7], t[8]])

    def stats(v):
        v = sorted(v)
        m = len(v)
        return {"n": m, "min": v[0], "p05": v[m // 20], "p25": v[m // 4], "median": v[m // 2], "mean": round(sum(v) / m, 1),
                "p75": v[(3 * m) // 4], "p95": v[(19 * m) // 20], "max": v[-1],
                "share_below_20.6us": round(sum(1 for a in v if a < 20600 * args.scale) / m, 4)}

    out_rows = []

    def emit(group, key, v):
        if len(v) >= 5:
            s = stats(v)
            out_rows.append([group, key] + [s[c] for c in ("n", "min", "p05", "p25", "median", "mean", "p75", "p95", "max", "share_below_20.6us")])

    phases = []
    for t in table:
        if t[1] not in phases:
            phases.append(t[1])
    for ph in phases:
        sel = [t for t in table if t[1] == ph]
        emit("phase", ph, [t[7] for t in sel])
        # skip each phase's first 50 launches below (ramp after a synchronize)
        body = [t for t in sel if t[2] >= 50]
        for slot in sorted({t[3] for t in body}):
            emit(f"{ph}:slot", slot, [t[7] for t in body if t[3] == slot])
        ys = sorted({t[5] for t in body})
        if len(ys) <= 16:
            for y in ys:
                emit(f"{ph}:y_addr", hex(y), [t[7] for t in body if t[5] == y])
        # the gap in front of the launch
        for lo, hi in ((-10**9, 800), (800, 1200), (1200, 1600), (1600, 2500), (2500, 10000), (10000, 10**12)):
            emit(f"{ph}:gap_ns[{max(lo, 0)},{hi if hi < 10**12 else 'inf'})", "", [t[7] for t in body if lo <= t[8] < hi])
        # position inside the phase (time since the phase began)
        if body:
            p0 = sel[0][6]
            for lo, hi in ((0, 5e3), (5e3, 2e4), (2e4, 1e5), (1e5, 1e6), (1e6, 1e9)):
                emit(f"{ph}:t_in_phase_us[{int(lo)},{int(hi)})", "", [t[7] for t in sel if lo <= t[6] - p0 < hi])
        # after a sync (i % 256 == 0) vs the rest
        emit(f"{ph}:first_after_queue_sync", "", [t[7] for t in sel if t[2] % 256 == 0])
    # duration histogram, 200 ns bins, per phase
    for ph in phases:
        sel = [t[7] for t in table if t[1] == ph]
        bins = {}
        for d in sel:
            b = int(d // (200 * args.scale)) * int(200 * args.scale)
            bins[b] = bins.get(b, 0) + 1
        for b in sorted(bins):
            out_rows.append([f"{ph}:hist_{int(200 * args.scale)}ns", b, bins[b]] + [""] * 9)
    # serial correlation: is a fast dispatch followed by a fast one?
    for ph in phases:
        d = [t[7] for t in table if t[1] == ph and t[2] >= 50]
        if len(d) > 100:
            m = sum(d) / len(d)
            num = sum((a - m) * (b - m) for a, b in zip(d, d[1:]))
            den = sum((a - m) ** 2 for a in d)
            out_rows.append([f"{ph}:lag1_autocorrelation", "", len(d), round(num / den, 4) if den else ""] + [""] * 8)
    with open(args.out + "_hist.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["group", "key", "n", "min_ns", "p05", "p25", "median", "mean", "p75", "p95", "max_ns", "share_fast"])
        w.writerow(["meta", json.dumps({k: log[k] for k in ("config", "dtype", "kernel", "ring", "bytes_per_launch", "build_id")})] + [""] * 10)
        w.writerows(out_rows)
    for r in out_rows:
        if r[0] == "phase":
            print(r)


def main():
    ap = argparse.ArgumentParser()
    sub = ap.add_subparsers(dest="cmd", required=True)
    r = sub.add_parser("run")
    r.add_argument("--config", default="cfg2")
    r.add_argument("--dtype", default="f32")
    r.add_argument("--launches", type=int, default=3000)
    r.add_argument("--log", required=True)
    a = sub.add_parser("analyze")
    a.add_argument("--log", required=True)
    a.add_argument("--trace", required=True, help="rocprofv3 output directory")
    a.add_argument("--out", required=True, help="output prefix")
    a.add_argument("--scale", type=float, default=1.0, help="scales the bin width and the 'fast' limit (0.5 for 64 MiB launches)")
    args = ap.parse_args()
    (run if args.cmd == "run" else analyze)(args)


if __name__ == "__main__":
    main()
