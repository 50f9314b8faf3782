"""Build libmctq_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m mct_quantizers_amd.hip.build [--force]

The library is written in-tree (mct_quantizers_amd/lib/) so it travels with the repository
snapshot to the GPU machine; it is git-ignored.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REPO = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
SOURCES = [os.path.join(CSRC, f) for f in ("mctq_misc.hip", "mctq_affine.hip", "mctq_codes.hip", "mctq_lut_scan.hip",
                                             "mctq_lut_table.hip", "mctq_grid.hip", "mctq_qlinear.hip", "mctq_codes4.hip", "mctq_codes_nhwc.hip")]
HEADERS = [os.path.join(REPO, "include", "mctq_hip.h"), os.path.join(CSRC, "mctq_kernels.hpp"),
           os.path.join(CSRC, "mctq_table_builder.h")]
OUT = os.path.join(PKG, "lib", "libmctq_hip.so")

# -ffp-contract=off / no fast-math: the kernels must reproduce IEEE float32 results bit for bit.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
         "-ffp-contract=off", "-fno-fast-math", "-Wall", "-Wno-unused-function"]


def needs_build() -> bool:
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(p) > t for p in SOURCES + HEADERS + [os.path.abspath(__file__)])


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return OUT
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found; cannot build libmctq_hip.so")
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    objdir = os.path.join(os.path.dirname(OUT), "obj")
    os.makedirs(objdir, exist_ok=True)
    inc = ["-I", os.path.join(REPO, "include"), "-I", CSRC]
    # one hipcc per translation unit, in parallel (the units are independent)
    procs = []
    for src in SOURCES:
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        cmd = [hipcc, *FLAGS, *inc, "-c", "-o", obj, src]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((obj, cmd, subprocess.Popen(cmd)))
    objs = []
    for obj, cmd, proc in procs:
        if proc.wait() != 0:
            raise subprocess.CalledProcessError(proc.returncode, cmd)
        objs.append(obj)
    tmp = OUT + ".tmp"
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp, *objs]
    if verbose:
        print(" ".join(link), flush=True)
    subprocess.run(link, check=True)
    os.replace(tmp, OUT)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(OUT)
