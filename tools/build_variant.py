#!/usr/bin/env python3
"""Experiment builds of libmctq_hip.so: the listed translation units recompiled with extra -D switches, and / or the
experiment translation units under tools/experiments/ added, everything else linked from the regular build's objects.
Output: tools/ablate/libmctq_hip_<NAME>.so (git-ignored, travels to the GPU box); run with
    MCTQ_HIP_LIB=tools/ablate/libmctq_hip_<NAME>.so MCTQ_BINDING=ctypes python tools/...
Usage: python tools/build_variant.py NAME [-DFLAG ...] [--units=a.hip,b.hip]
       python tools/build_variant.py lut_compact          (regular objects + tools/experiments/lut_compact/mctq_lut_compact.hip)
       python tools/build_variant.py lut_conflicts | rowsteps_sched   (regular objects + tools/experiments/<name>/<name>.hip)"""
import os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from mct_quantizers_amd.hip import build as B

def main():
    name, defs, units = sys.argv[1], [a for a in sys.argv[2:] if a.startswith("-D")], ["mctq_lut_table.hip"]
    extra = []
    if name == "lut_compact":
        units, xdir = [], os.path.join(REPO, "tools", "experiments", "lut_compact")
        extra = [(os.path.join(xdir, "mctq_lut_compact.hip"), ["-I", xdir])]
    elif name in ("lut_conflicts", "rowsteps_sched", "chanlast2"):
        units, xdir = [], os.path.join(REPO, "tools", "experiments", name)
        extra = [(os.path.join(xdir, name + ".hip"), ["-I", xdir])]
    for a in sys.argv[2:]:
        if a.startswith("--units"):
            units = a.split("=", 1)[1].split(",")
    B.build()
    objdir = os.path.join(os.path.dirname(B.OUT), "obj")
    out_dir = os.path.join(REPO, "tools", "ablate")
    os.makedirs(os.path.join(out_dir, "obj_" + name), exist_ok=True)
    bid = B.tree_build_id()
    objs, procs = [], []
    for src in B.SOURCES:
        base = os.path.basename(src)
        stamp = ['-DMCTQ_BUILD_ID="%s"' % bid] if base == "mctq_misc.hip" else []
        if base in units:
            obj = os.path.join(out_dir, "obj_" + name, base + ".o")
            cmd = ["hipcc", *B.FLAGS, *defs, "-I", os.path.join(REPO, "include"), "-I", B.CSRC, "-c", "-o", obj, src]
            print(" ".join(cmd), flush=True)
            procs.append(subprocess.Popen(cmd))
        else:
            obj = os.path.join(objdir, f"{base}.{B._digest([src] + B.HEADERS, B.FLAGS + stamp)}.o")
            assert os.path.exists(obj), obj
        objs.append(obj)
    for src, inc in extra:
        obj = os.path.join(out_dir, "obj_" + name, os.path.basename(src) + ".o")
        cmd = ["hipcc", *B.FLAGS, *defs, "-I", os.path.join(REPO, "include"), "-I", B.CSRC, *inc, "-c", "-o", obj, src]
        print(" ".join(cmd), flush=True)
        procs.append(subprocess.Popen(cmd))
        objs.append(obj)
    assert all(p.wait() == 0 for p in procs)
    out = os.path.join(out_dir, f"libmctq_hip_{name}.so")
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, *objs], check=True)
    print(out)

main()
