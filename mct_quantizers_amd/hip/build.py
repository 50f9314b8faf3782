"""Build libmctq_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m mct_quantizers_amd.hip.build [--force]

The library is written in-tree (mct_quantizers_amd/lib/) so it travels with the repository
snapshot to the GPU machine; it is git-ignored.
"""
from __future__ import annotations

import contextlib
import fcntl
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REPO = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
SOURCES = [os.path.join(CSRC, f) for f in ("mctq_misc.hip", "mctq_affine.hip", "mctq_codes.hip", "mctq_lut_scan.hip",
                                             "mctq_lut_table.hip", "mctq_grid.hip", "mctq_qlinear.hip", "mctq_codes4.hip", "mctq_codes_nhwc.hip", "mctq_batched.hip", "mctq_f64.hip", "mctq_lut_steps.hip")]
HEADERS = [os.path.join(REPO, "include", "mctq_hip.h"), os.path.join(CSRC, "mctq_kernels.hpp"),
           os.path.join(CSRC, "mctq_table_builder.h")]
OUT = os.path.join(PKG, "lib", "libmctq_hip.so")
BINDING_SRC = os.path.join(CSRC, "binding", "mctq_torch.cpp")
BINDING_OUT = os.path.join(PKG, "lib", "_mctq_torch.so")

# -ffp-contract=off / no fast-math: the kernels must reproduce IEEE float32 results bit for bit.
# -amdgpu-kernarg-preload-count: leading scalar kernel arguments arrive in SGPRs at wave start (gfx940+); kernels
# that take a struct first are unaffected.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
         "-ffp-contract=off", "-fno-fast-math", "-Wall", "-Wno-unused-function",
         "-mllvm", "-amdgpu-kernarg-preload-count=16"]


@contextlib.contextmanager
def _build_lock():
    """One builder at a time per checkout (several ranks or test workers may call build() at once): the others wait,
    then find the library up to date."""
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    with open(os.path.join(os.path.dirname(OUT), ".build.lock"), "w") as f:
        fcntl.flock(f, fcntl.LOCK_EX)
        try:
            yield
        finally:
            fcntl.flock(f, fcntl.LOCK_UN)


def needs_build() -> bool:
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(p) > t for p in SOURCES + HEADERS + [os.path.abspath(__file__)])


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return OUT
    with _build_lock():
        if not force and not needs_build():          # another process built it while this one waited
            return OUT
        return _build(verbose)


def _build(verbose: bool) -> str:
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


def binding_needs_build() -> bool:
    if not os.path.exists(BINDING_OUT):
        return True
    t = os.path.getmtime(BINDING_OUT)
    return any(os.path.getmtime(p) > t for p in (BINDING_SRC, HEADERS[0], OUT, os.path.abspath(__file__)))


def build_binding(force: bool = False, verbose: bool = True) -> str:
    """Compile the CPython binding of the hot entry points (host-only C++: g++, no device code).

    It links libtorch (tensor type, caching allocator, current stream) and libmctq_hip.so; both are found at run
    time through rpaths ($ORIGIN for the kernels' library, torch's own lib directory)."""
    if not force and not binding_needs_build():
        return BINDING_OUT
    with _build_lock():
        if not force and not binding_needs_build():
            return BINDING_OUT
        return _build_binding(verbose)


def _build_binding(verbose: bool) -> str:
    import sysconfig
    import torch
    tdir = os.path.dirname(torch.__file__)
    cxx = shutil.which("g++") or shutil.which("c++")
    if cxx is None:
        raise RuntimeError("g++ not found; cannot build the compiled binding")
    abi = int(getattr(torch._C, "_GLIBCXX_USE_CXX11_ABI", True))
    tmp = BINDING_OUT + ".tmp"
    cmd = [cxx, "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function",
           "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", f"-D_GLIBCXX_USE_CXX11_ABI={abi}",
           "-I", os.path.join(REPO, "include"), "-I", os.path.join(tdir, "include"),
           "-I", os.path.join(tdir, "include", "torch", "csrc", "api", "include"), "-I", "/opt/rocm/include",
           "-I", sysconfig.get_paths()["include"], BINDING_SRC, "-o", tmp,
           "-L", os.path.join(tdir, "lib"), "-L", os.path.dirname(OUT),
           "-ltorch", "-ltorch_cpu", "-ltorch_hip", "-lc10", "-lc10_hip", "-ltorch_python", "-lmctq_hip",
           "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + os.path.join(tdir, "lib")]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    os.replace(tmp, BINDING_OUT)
    return BINDING_OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(OUT)
    build_binding(force="--force" in sys.argv)
    print(BINDING_OUT)
