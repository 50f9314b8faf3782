"""Build libmctq_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m mct_quantizers_amd.hip.build [--force]

The library is written in-tree (mct_quantizers_amd/lib/) so it travels with the repository
snapshot to the GPU machine; it is git-ignored.
"""
from __future__ import annotations

import contextlib
import fcntl
import hashlib
import mmap
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REPO = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
# slowest translation units first (they are started in this order; the build's critical path is mctq_batched_lut.hip, ~35 s)
SOURCES = [os.path.join(CSRC, f) for f in ("mctq_batched_lut.hip", "mctq_lut_steps.hip", "mctq_lut_table.hip", "mctq_lut_scan.hip",
                                             "mctq_qlinear.hip", "mctq_batched.hip", "mctq_affine.hip", "mctq_codes.hip",
                                             "mctq_f64.hip", "mctq_grid.hip", "mctq_codes4.hip", "mctq_codes_nhwc.hip",
                                             "mctq_misc.hip")]
HEADERS = [os.path.join(REPO, "include", "mctq_hip.h"), os.path.join(CSRC, "mctq_kernels.hpp"),
           os.path.join(CSRC, "mctq_table_builder.h"), os.path.join(CSRC, "mctq_batched.hpp")]
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


# ---- what a binary was built from: a content hash, not modification times ----------------------------------------------
# The built .so files are git-ignored but travel to the GPU machine with the snapshot; a binary that is NEWER than the
# sources yet built from different ones must not be used silently.  The library carries the hash of every source, header
# and flag it was compiled from ("MCTQ_BUILD_ID=<id>" in its read-only data, also returned by mctq_build_id()); the tree's
# hash is recomputed from the files (~1 ms) and compared -- by needs_build() before deciding to reuse, and by the loader
# (hip/native.py) before any call.

UNIT_SECONDS = {}            # translation unit -> compile wall seconds of the last _build() in this process (None: cached object reused)

ID_MARKER = b"MCTQ_BUILD_ID="
BINDING_ID_MARKER = b"MCTQ_BINDING_ID="


def _digest(paths, extra=()) -> str:
    h = hashlib.sha256()
    for p in paths:
        h.update(os.path.basename(p).encode() + b"\0")
        with open(p, "rb") as f:
            h.update(f.read())
        h.update(b"\0")
    for e in extra:
        h.update(str(e).encode() + b"\0")
    return h.hexdigest()[:16]


def tree_build_id() -> str:
    """Hash of the kernel sources, the headers and the compiler flags as they are in the tree now."""
    return _digest(sorted(SOURCES) + HEADERS, FLAGS)


def embedded_id(path: str, marker: bytes = ID_MARKER):
    """The id stamped into a built binary, read from the file without loading it; None if absent."""
    try:
        with open(path, "rb") as f, mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ) as m:
            at = m.find(marker)
            while at >= 0:
                tail = m[at + len(marker):at + len(marker) + 16]
                if len(tail) == 16 and all(c in b"0123456789abcdef" for c in tail):
                    return tail.decode()
                at = m.find(marker, at + 1)
    except (OSError, ValueError):
        pass
    return None


def needs_build() -> bool:
    return embedded_id(OUT) != tree_build_id()


def build(force: bool = False, verbose: bool = True) -> str:
    """Build the library unless the one in lib/ carries the tree's id.  ``force``: recompile every translation unit
    (the per-unit object cache is ignored too)."""
    if not force and not needs_build():
        return OUT
    with _build_lock():
        if not force and not needs_build():          # another process built it while this one waited
            return OUT
        return _build(verbose, force)


def _build(verbose: bool, force: bool = False) -> str:
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found; cannot build libmctq_hip.so")
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    objdir = os.path.join(os.path.dirname(OUT), "obj")
    os.makedirs(objdir, exist_ok=True)
    inc = ["-I", os.path.join(REPO, "include"), "-I", CSRC]
    build_id = tree_build_id()
    # one hipcc per translation unit, in parallel (the units are independent).  Objects are cached by the hash of what
    # they were compiled from (the unit, every header, the flags): an unchanged unit is not recompiled.
    import time
    procs, objs, keep = [], [], set()
    UNIT_SECONDS.clear()
    for src in SOURCES:
        stamp = ['-DMCTQ_BUILD_ID="%s"' % build_id] if os.path.basename(src) == "mctq_misc.hip" else []
        key = _digest([src] + HEADERS, FLAGS + stamp)
        obj = os.path.join(objdir, f"{os.path.basename(src)}.{key}.o")
        objs.append(obj)
        keep.add(os.path.basename(obj))
        if os.path.exists(obj) and not force:
            UNIT_SECONDS[os.path.basename(src)] = None          # object of the same sources, headers and flags reused
            continue
        cmd = [hipcc, *FLAGS, *stamp, *inc, "-c", "-o", obj + ".tmp", src]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((obj, cmd, subprocess.Popen(cmd), time.time(), os.path.basename(src)))
    # wall time per translation unit (the jobs run in parallel: a unit's time is start -> exit as seen by polling)
    pending = list(procs)
    while pending:
        for job in list(pending):
            obj, cmd, proc, t0, name = job
            rc = proc.poll()
            if rc is None:
                continue
            pending.remove(job)
            UNIT_SECONDS[name] = time.time() - t0
            if rc != 0:
                for other in pending:
                    other[2].kill()
                raise subprocess.CalledProcessError(rc, cmd)
            os.replace(obj + ".tmp", obj)
        if pending:
            time.sleep(0.05)
    keep.add("mctq_torch.binding.o")                 # the binding's object (build_all compiles it beside these jobs)
    for name in os.listdir(objdir):                  # objects of earlier source versions
        if name not in keep and not name.endswith(".tmp"):
            os.remove(os.path.join(objdir, name))
    tmp = OUT + ".tmp"
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp, *objs]
    if verbose:
        print(" ".join(link), flush=True)
    subprocess.run(link, check=True)
    if embedded_id(tmp) != build_id:
        raise RuntimeError(f"{tmp} does not carry build id {build_id}")
    os.replace(tmp, OUT)
    return OUT


def _binding_flags():
    import torch
    abi = int(getattr(torch._C, "_GLIBCXX_USE_CXX11_ABI", True))
    return ["-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function",
            "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", f"-D_GLIBCXX_USE_CXX11_ABI={abi}"]


def binding_build_id() -> str:
    """Hash of the binding's source, the C header, its flags, the torch version it links and the library's id."""
    import torch
    return _digest([BINDING_SRC, HEADERS[0]], _binding_flags() + [torch.__version__, tree_build_id()])


def binding_needs_build() -> bool:
    return embedded_id(BINDING_OUT, BINDING_ID_MARKER) != binding_build_id()


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


def _binding_compile_cmd(obj: str):
    """g++ -c of the binding (the slow part: torch's headers); needs no library, so it can run beside the hipcc jobs."""
    import sysconfig
    import torch
    tdir = os.path.dirname(torch.__file__)
    cxx = shutil.which("g++") or shutil.which("c++")
    if cxx is None:
        raise RuntimeError("g++ not found; cannot build the compiled binding")
    flags = [f for f in _binding_flags() if f != "-shared"]
    return [cxx, *flags, '-DMCTQ_BINDING_ID="%s"' % binding_build_id(),
            "-I", os.path.join(REPO, "include"), "-I", os.path.join(tdir, "include"),
            "-I", os.path.join(tdir, "include", "torch", "csrc", "api", "include"), "-I", "/opt/rocm/include",
            "-I", sysconfig.get_paths()["include"], "-c", BINDING_SRC, "-o", obj]


def _binding_link(obj: str, verbose: bool) -> str:
    import torch
    tdir = os.path.dirname(torch.__file__)
    cxx = shutil.which("g++") or shutil.which("c++")
    tmp = BINDING_OUT + ".tmp"
    cmd = [cxx, "-shared", "-fPIC", obj, "-o", tmp, "-L", os.path.join(tdir, "lib"), "-L", os.path.dirname(OUT),
           "-ltorch", "-ltorch_cpu", "-ltorch_hip", "-lc10", "-lc10_hip", "-ltorch_python", "-lmctq_hip",
           "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + os.path.join(tdir, "lib")]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    bid = binding_build_id()
    if embedded_id(tmp, BINDING_ID_MARKER) != bid:
        raise RuntimeError(f"{tmp} does not carry binding id {bid}")
    os.replace(tmp, BINDING_OUT)
    return BINDING_OUT


def _build_binding(verbose: bool) -> str:
    objdir = os.path.join(os.path.dirname(OUT), "obj")
    os.makedirs(objdir, exist_ok=True)
    obj = os.path.join(objdir, "mctq_torch.binding.o")
    cmd = _binding_compile_cmd(obj)
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return _binding_link(obj, verbose)


def build_all(force: bool = False, verbose: bool = True):
    """Library + binding, the binding's compile step running beside the hipcc jobs (a clean build is then as long as the
    slowest translation unit + two link steps: ~45 s on 8 cores).  Returns (library path, binding path, seconds)."""
    import time
    t0 = time.time()
    with _build_lock():
        lib_needed = force or needs_build()
        bind_needed = force or binding_needs_build()
        proc = obj = None
        if bind_needed:
            objdir = os.path.join(os.path.dirname(OUT), "obj")
            os.makedirs(objdir, exist_ok=True)
            obj = os.path.join(objdir, "mctq_torch.binding.o")
            cmd = _binding_compile_cmd(obj)
            if verbose:
                print(" ".join(cmd), flush=True)
            proc = subprocess.Popen(cmd)
        try:
            if lib_needed:
                _build(verbose, force)
        finally:
            rc = proc.wait() if proc is not None else 0
        if rc != 0:
            raise subprocess.CalledProcessError(rc, "g++ (binding)")
        if bind_needed:
            _binding_link(obj, verbose)
    return OUT, BINDING_OUT, time.time() - t0


if __name__ == "__main__":
    lib, binding, seconds = build_all(force="--force" in sys.argv)
    print(lib)
    print(binding)
    print(f"{seconds:.1f} s")
