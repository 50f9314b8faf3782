#!/usr/bin/env python3
"""Kernel time vs launch gaps from a rocprofv3 --kernel-trace CSV: for the dominant kernel, the duration of each
dispatch and the idle time on the GPU between the end of one dispatch and the start of the next.
usage: trace_gaps.py <kernel_trace.csv> <label>"""
import csv, statistics, sys
rows = list(csv.DictReader(open(sys.argv[1])))
by = {}
for r in rows:
    by.setdefault(r["Kernel_Name"], []).append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
name = max(by, key=lambda k: len(by[k]))
d = sorted(by[name])
d = d[len(d) // 4:]                       # drop warm-up / pre-warm part of the run: keep the last three quarters
dur = [e - s for s, e in d]
gap = [d[i + 1][0] - d[i][1] for i in range(len(d) - 1)]
per = [d[i + 1][0] - d[i][0] for i in range(len(d) - 1)]
q = lambda v, p: sorted(v)[int(len(v) * p)]
print(f"{sys.argv[2]}: {name[:110]}")
print(f"    dispatches {len(d)}   duration ns: median {statistics.median(dur):.0f} mean {statistics.mean(dur):.0f} p10 {q(dur, .1)} p90 {q(dur, .9)}")
print(f"    gap end->next start ns: median {statistics.median(gap):.0f} mean {statistics.mean(gap):.0f} p10 {q(gap, .1)} p90 {q(gap, .9)}")
print(f"    period start->start ns: median {statistics.median(per):.0f} mean {statistics.mean(per):.0f}   "
      f"=> GPU busy {100.0 * statistics.mean(dur) / statistics.mean(per):.0f} % of the period; the rest is launch gap (host-bound when the gap >> 1.4 us)")
