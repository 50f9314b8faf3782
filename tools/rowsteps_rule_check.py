#!/usr/bin/env python3
"""Does the dispatcher's window for rowsteps_kernel (csrc/mctq_kernels.hpp: grid of four-step blocks = one round of the 8
resident blocks per CU) pick the kernel that measured faster?  Reads tools/rowsteps_probe.py's log (30 shapes x storage types,
both kernels forced through the tuning key) and applies the rule to every line.
    python tools/rowsteps_rule_check.py profiles/r05/rowsteps_probe_sustained.log"""
import re
import sys

CUS = 256
rows = []
for ln in open(sys.argv[1]):
    m = re.match(r"(\w+) (\d+)x(\d+) \((\d+) MiB per launch.*?rowsteps=0:\s+([\d.]+) us.*?rowsteps=1:\s+([\d.]+) us", ln)
    if not m:
        continue
    dt, r, c, mib, t0, t1 = m.group(1), int(m.group(2)), int(m.group(3)), int(m.group(4)), float(m.group(5)), float(m.group(6))
    spr = c // ((4 if dt == "f32" else 8) * 256)                 # 256-lane-vector steps per row
    blocks, rnd = (r * spr + 3) // 4, 8 * CUS
    rows.append((dt, r, c, mib, spr, blocks / rnd, t0, t1, blocks <= rnd and blocks * 4 >= rnd * 3))
print(f"{'storage':7s} {'shape':12s} {'MiB':>4s} {'steps/row':>9s} {'rounds':>6s} {'rows_kernel':>11s} {'rowsteps':>9s} {'gain %':>7s}  window")
miss = 0
for x in sorted(rows, key=lambda x: (x[4], x[5])):
    gain = (x[6] - x[7]) / x[6] * 100
    wrong = abs(gain) > 1 and (gain > 1) != x[8]
    miss += wrong
    print(f"{x[0]:7s} {str(x[1]) + 'x' + str(x[2]):12s} {x[3]:4d} {x[4]:9d} {x[5]:6.2f} {x[6]:11.2f} {x[7]:9.2f} {gain:7.1f}  {'yes' if x[8] else 'no '}"
          f"{'   <-- the rule picks the slower kernel' if wrong else ''}")
print(f"{len(rows)} cases; the rule picks the slower kernel (by more than 1 %) in {miss}")
