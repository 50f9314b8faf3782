"""The C-ABI library loads here (no GPU) and exports every symbol include/mctq_hip.h declares."""
import ctypes
import os
import re

import pytest

from conftest import REPO


def _declared_symbols():
    text = open(os.path.join(REPO, "include", "mctq_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mctq_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    from mct_quantizers_amd.hip import native
    assert _declared_symbols() == sorted(native.SIGNATURES)


def test_library_loads_and_exports_everything():
    from mct_quantizers_amd.hip import build, native
    path = build.build()                        # hipcc cross-compiles for gfx950 without a GPU
    lib = ctypes.CDLL(path)
    for name in _declared_symbols():
        assert hasattr(lib, name), name
    handle = native.load()
    assert handle.mctq_abi_version() == native.ABI_VERSION
    assert isinstance(handle.mctq_last_error(), bytes)       # "" until a call fails on this thread (tests share one)


def test_argument_validation_needs_no_gpu():
    from mct_quantizers_amd.hip import native
    lib = native.load()
    assert lib.mctq_fq_per_tensor_f32(None, None, -5, 1.0, 0, 0, 255, None) == native.MCTQ_E_ARG
    assert b"n < 0" in lib.mctq_last_error()
    assert lib.mctq_fq_per_tensor_f32(None, None, 0, 1.0, 0, 0, 255, None) == 0          # empty: nothing launched
    assert lib.mctq_fq_per_channel_f32(None, None, 0, 4, 8, None, None, 0, 255, None) == 0
    assert lib.mctq_lut_per_tensor_f32(None, None, 0, 1.0, 1.0, None, 4, 128.0, -128.0, 127.0, None) \
        == native.MCTQ_E_ARG                                                              # lut NULL
    assert lib.mctq_set_tuning(b"unroll", 3) == native.MCTQ_E_ARG
    assert lib.mctq_set_tuning(b"unroll", 4) == 0


def test_missing_library_is_loud(monkeypatch):
    import pytest
    from mct_quantizers_amd.hip import native
    monkeypatch.setattr(native, "_lib", None)
    monkeypatch.setenv("MCTQ_HIP_LIB", "/nonexistent/libmctq_hip.so")
    with pytest.raises(native.NativeLibraryError):
        native.load()
    assert not native.is_available()


def test_decision_table_builder_matches_the_oracle_scan():
    """Host-only: the LDS decision table reproduces the literal first-minimum scan around every boundary."""
    import numpy as np
    from mct_quantizers_amd.hip import native
    from oracle import mctq_oracle as O
    rng = np.random.default_rng(0)
    luts = [[-5.0, 5.0], [3.0, 3.0, -8.0], [22.0, -53.0, 62.0, 0.0, -66.0, -21.0, 44.0, -40.0],
            [float(v) for v in rng.permutation(np.arange(-128, 128))], [7.0], [-1.0, 0.0, 1.0, 2.0, 3.0]]
    for lut in luts:
        tab = native.build_lut_table(lut, 128.0, -128.0, 127.0)
        assert tab is not None and tab.shape == (512, 2)
        k = rng.integers(-256, 255, size=100000).astype(np.float32) * np.float32(0.5)
        off = rng.integers(-40, 41, size=k.size).astype(np.int64)
        b = k.view(np.int32).astype(np.int64)
        b = np.where(k > 0, b + off, np.where(k < 0, b - off, b))          # walk +-40 ulps around each point
        t = np.concatenate([b.astype(np.int32).view(np.float32), rng.uniform(-128, 127, 50000).astype(np.float32)])
        t = np.clip(t, -128, 127).astype(np.float32)
        want = O.lut_quantize(t, lut, np.asarray([128.0], np.float32), True, 8, 0.0)   # thr=128, eps=0: t == x
        e = tab[(t * np.float32(2) + np.float32(256.5)).astype(np.int32)]
        halves = np.ascontiguousarray(e[:, 1]).view(np.float16).reshape(-1, 2).astype(np.float32)   # (below, above)
        got = np.where(t >= e[:, 0], halves[:, 1], halves[:, 0]) * np.float32(128.0)
        assert np.array_equal(got, want)
        assert tab[511, 0] == np.float32(lut[0]) / np.float32(128.0)        # NaN input -> codebook entry 0
    assert native.build_lut_table([0.5, 1.0], 128.0, -128.0, 127.0) is None          # non-integer codebook
    assert native.build_lut_table([1.0], 2.0 ** 12, -2048.0, 2047.0) is None         # too wide for LDS


def test_threshold_list_builder_matches_the_oracle_scan():
    """Host-only: the sorted threshold list (wide integer codebooks, lut_values_bitwidth > 10) reproduces the literal
    first-minimum scan: +-60 ulps around every threshold, at every half-integer of a window, and on random points."""
    import numpy as np
    from mct_quantizers_amd.hip import native
    from oracle import mctq_oracle as O
    rng = np.random.default_rng(1)
    cases = [([-5.0, 5.0], True, 12), ([3.0, 3.0, -8.0], True, 12), ([7.0], True, 12),
             ([float(v) for v in rng.choice(np.arange(-2048, 2048), 16, replace=False)], True, 12),
             ([float(v) for v in rng.choice(np.arange(-32768, 32768), 256, replace=False)], True, 16),
             ([float(v) for v in rng.choice(np.arange(0, 4097), 64, replace=False)], False, 12),
             ([float(v) for v in rng.choice(np.arange(-32768, 32768), 1024, replace=False)], True, 16),
             ([float(v) for v in rng.permutation(np.arange(-128, 128))], True, 8)]
    for lut, signed, B in cases:
        mult = float(2 ** (B - int(signed)))
        cmin, cmax = (float(-2 ** (B - 1)), float(2 ** (B - 1) - 1)) if signed else (0.0, float(2 ** B - 1))
        st = native.build_lut_steps(lut, mult, cmin, cmax)
        assert st is not None
        P = 1
        while P < len(set(lut)):
            P *= 2
        G_for = 0 if P < 128 else (1024 if P <= 256 else (4096 if P <= 1024 else 8192))
        assert int(st[2 * P + 1]) == P and st.shape[0] in (2 * P + 2, 2 * P + 2 + 4 + G_for)
        has_cells = st.shape[0] != 2 * P + 2
        assert has_cells == (P >= 128)
        T, Q = st[:P], st[P:2 * P]
        assert np.all(np.diff(T[1:]) >= 0) or P <= 2
        fin = T[1:][np.isfinite(T[1:])]
        b = fin.view(np.int32).astype(np.int64)[:, None]
        off = np.arange(-60, 61, dtype=np.int64)[None, :]
        near = np.where(fin[:, None] > 0, b + off, b - off).astype(np.int32).view(np.float32).reshape(-1)
        halves = (rng.integers(int(2 * cmin), int(2 * cmax) + 1, size=20000) * 0.5).astype(np.float32)
        t = np.concatenate([near, halves, rng.uniform(cmin, cmax, 50000).astype(np.float32)])
        t = np.clip(t, cmin, cmax).astype(np.float32)
        want = O.lut_quantize(t, lut, np.asarray([mult], np.float32), signed, B, 0.0)      # thr = mult, eps = 0: t == x
        idx = np.searchsorted(T[1:], t, side="right")                                      # thresholds <= t
        got = Q[idx] * np.float32(mult)
        assert np.array_equal(got, want), (lut[:8], B)
        if has_cells:                                                                      # the kernel's cell-index route
            G, maxc, gscale, c0 = int(st[2 * P + 2]), int(st[2 * P + 3]), np.float32(st[2 * P + 4]), np.float32(st[2 * P + 5])
            cells = st[2 * P + 6: 2 * P + 6 + G].view(np.uint32)
            assert G == G_for and 0 < maxc <= 4 and c0 == np.float32(cmin)
            v = ((t - c0).astype(np.float32) * gscale).astype(np.float32)
            c = np.clip(v, 0, G - 1).astype(np.int32)
            first, n = (cells[c] & 0xffff).astype(np.int64), (cells[c] >> 16).astype(np.int64)
            idx2 = first.copy()
            for j in range(maxc):
                th = T[np.minimum(1 + first + j, P - 1)]
                idx2 += ((j < n) & (t >= th)).astype(np.int64)
            assert np.array_equal(idx2, idx), (B, len(lut))
        assert st[2 * P] == np.float32(lut[0]) / np.float32(mult)                          # NaN input -> codebook entry 0
    assert native.build_lut_steps([0.5, 1.0], 2048.0, -2048.0, 2047.0) is None             # non-integer codebook


def test_table_builder_under_address_and_ub_sanitizers(tmp_path):
    """The host half of the library (decision-table construction) is plain C++: build it alone with
    AddressSanitizer + UBSan and run its self-check (GPU sanitizers are unavailable on this pool)."""
    import shutil
    import subprocess
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("g++ not found")
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "table_builder_check")
    subprocess.run([gxx, "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                    "-I", os.path.join(repo, "mct_quantizers_amd", "csrc"), "-o", exe,
                    os.path.join(repo, "tests", "native", "table_builder_check.cpp")], check=True)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "table builder ok" in out.stdout


def test_compiled_binding_builds_loads_and_declines_cpu_tensors():
    """The CPython binding of the hot entry points (csrc/binding/mctq_torch.cpp): builds with g++ here, loads without a
    GPU, exports every callable the Python side uses, reports the library's ABI version, and answers NotImplemented
    (never a wrong result, never a crash) for anything that is not a plain eager HIP tensor."""
    import importlib.util
    import torch
    from mct_quantizers_amd.hip import build, native
    build.build()
    path = build.build_binding()
    assert os.path.exists(path)
    native.load()
    spec = importlib.util.spec_from_file_location(native.FAST_NAME, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.abi_version() == native.ABI_VERSION
    for name in ("fq_per_tensor", "fq_per_channel", "fq_per_tensor_tqp", "lutt_per_tensor", "lutt_per_channel", "fq_batched",
                 "AffinePlan", "LutPlan", "BatchPlan"):
        assert callable(getattr(mod, name)), name
    x = torch.randn(4, 8)
    s, z = torch.ones(4), torch.zeros(4, dtype=torch.int32)
    assert mod.fq_per_tensor(x, 0.1, 0, -8, 7) is NotImplemented
    assert mod.fq_per_channel(x, s, z, 0, -8, 7) is NotImplemented
    assert mod.fq_per_tensor_tqp(x, s[:1], z[:1], -8, 7) is NotImplemented
    assert mod.fq_batched([(x, s, None, 0, -8, 7)]) is NotImplemented
    assert mod.fq_batched([]) == []
    assert mod.AffinePlan(0.1, 0, -8, 7)(x) is NotImplemented
    assert mod.AffinePlan(s, None, 0, -8, 7)(x) is NotImplemented
    assert mod.LutPlan(torch.zeros(5, 2), 1.0, 1.0, 1.0, 1.0, 128.0, -128.0, 127.0, 1)(x) is NotImplemented
    assert mod.fq_per_tensor("not a tensor", 0.1, 0, -8, 7) is NotImplemented
    with pytest.raises(TypeError):
        mod.fq_per_tensor(x, 0.1, 0, -8)                      # wrong arity
    with pytest.raises(TypeError):
        mod.BatchPlan([(x, x, s, None, 0, -8, 7)])            # a plan needs HIP tensors
    with pytest.raises(TypeError):
        mod.AffinePlan(0.1, 0)


def test_a_stale_binary_is_rebuilt_whatever_its_modification_time(tmp_path, monkeypatch):
    """The build decision is a content hash (hip/build.py): a source whose text changed while its mtime did not -- or a
    binary that is newer than sources it was not built from -- must count as stale, and the loader must refuse it."""
    import shutil
    from mct_quantizers_amd.hip import build, native
    build.build()
    assert not build.needs_build()
    assert build.embedded_id(build.OUT) == build.tree_build_id() == native.load().mctq_build_id().decode()
    src = build.SOURCES[0]
    copy = str(tmp_path / os.path.basename(src))
    shutil.copy2(src, copy)
    st = os.stat(copy)
    with open(copy, "a") as f:
        f.write("\n// a kernel constant edited\n")
    os.utime(copy, (st.st_atime, st.st_mtime))                 # the modification time is what it was
    assert os.stat(copy).st_mtime == st.st_mtime
    monkeypatch.setattr(build, "SOURCES", [copy] + build.SOURCES[1:])
    assert build.tree_build_id() != build.embedded_id(build.OUT)
    assert build.needs_build() and build.binding_needs_build()  # the binding's id covers the library's
    with pytest.raises(native.NativeLibraryError, match="built from other sources"):
        native._check_build_id(build.OUT, build.embedded_id(build.OUT), "tree_build_id")
    monkeypatch.undo()
    assert not build.needs_build()
    # a flag change is a source change too
    monkeypatch.setattr(build, "FLAGS", build.FLAGS + ["-DX=1"])
    assert build.needs_build()


def test_rowsteps_kernel_keeps_the_schedule_its_dispatch_window_was_measured_with(tmp_path):
    """rowsteps_kernel (csrc/mctq_kernels.hpp) is taken where its grid is one round of resident blocks because it measured
    6-15 % faster there than rows_kernel (profiles/r05/rowsteps_sched.log: 4 of 4 shapes in the window, three boxes) -- and what
    makes it fast is a property of the COMPILED code, not of its source: the block's first three 16-byte loads are issued
    before any of the parameter fetch (row arithmetic, scalar loads, the IEEE reciprocal), the fourth once the first has
    landed.  Nine source forms that try to WRITE that schedule compile to something slower (LLVM commons the row division
    above the loads, or sinks the loads to their first use; tools/experiments/rowsteps_sched/).  So the schedule is ASSERTED
    here, on the code hipcc generates from the shipped source with the shipped flags: if a compiler or source change moves
    it, this test fails and the window (tuning key "rowsteps", default 2) has to be re-measured instead of silently losing
    its reason."""
    import shutil
    import subprocess
    from mct_quantizers_amd.hip import build as B
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    asm = tmp_path / "affine.s"
    subprocess.run([hipcc, *B.FLAGS, "-I", os.path.join(REPO, "include"), "-I", B.CSRC, "--cuda-device-only", "-S", "-o", str(asm),
                    os.path.join(B.CSRC, "mctq_affine.hip")], check=True, capture_output=True)
    text = asm.read_text()
    names = re.findall(r"^(_ZN4mctq15rowsteps_kernel\w+):", text, flags=re.M)
    assert len(names) == 6, names                             # float32 / float16 / bfloat16 x two cache policies
    for name in names:
        body = text[text.index(name + ":"):]
        body = body[:body.index("s_endpgm")]
        ins = [ln.split(";")[0].strip() for ln in body.splitlines()]
        ins = [ln for ln in ins if ln and not ln.startswith(".") or re.match(r"\.LBB\d+_\d+:", ln)]
        # basic blocks of the function, in layout order
        blocks, cur = [], []
        for ln in ins:
            if re.match(r"\.LBB\d+_\d+:", ln):
                blocks.append(cur)
                cur = []
                continue
            cur.append(ln)
            if ln.startswith(("s_cbranch", "s_branch")):
                blocks.append(cur)
                cur = []
        blocks.append(cur)
        # the hot (all four steps present) path: the first block that issues three data loads in a row
        hot = next((i for i, b in enumerate(blocks) if sum(x.startswith("global_load_dwordx4") for x in b) >= 3), None)
        assert hot is not None, f"{name}: no block with three back-to-back data loads"
        before = [x for b in blocks[:hot] for x in b] + blocks[hot][:max(i for i, x in enumerate(blocks[hot]) if x.startswith("global_load_dwordx4"))]
        # nothing of the parameter fetch in front of them: no scalar load of a table element, no reciprocal
        def is_fetch(x):                                      # a table element by scalar load (kernel arguments come from s[0:1]), a reciprocal
            return bool(re.match(r"s_load_dword s\d+, s\[(?!0:1\])", x)) or x.startswith(("v_rcp_f32", "v_div_fixup_f32"))
        assert not [x for x in before if is_fetch(x)], f"{name}: parameter fetch scheduled in front of the data loads"
        after = [x for b in blocks[hot:] for x in b]
        loads = [i for i, x in enumerate(after) if x.startswith("global_load_dwordx4")]
        assert len(loads) >= 4
        between = after[loads[2] + 1:loads[3]]
        assert not [x for x in between if is_fetch(x)], f"{name}: parameter fetch in front of the fourth data load"
        if "Eff" not in name:
            # 16-bit storage: the fourth load waits for the first one (at most two loads outstanding in front of it)
            assert any(re.match(r"s_waitcnt vmcnt\([012]\)", x) for x in between), f"{name}: the fourth load is not staggered"
        # the parameter fetch (scalar loads, IEEE reciprocal) runs under the loads' latency
        assert any(is_fetch(x) for x in after[loads[3]:]), name


# ---- the headline kernel's generated code ---------------------------------------------------------------

HEADLINE = "_ZN4mctq11rows_kernelINS_8AffineOpEffLi4ELi1EEEvPKT0_PT1_jjjT_"       # rows_kernel<AffineOp, float, float, 4, 1>


def _compile_affine_asm(tmp_path, src_dir=None):
    import shutil
    import subprocess
    from mct_quantizers_amd.hip import build as B
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    asm = tmp_path / f"affine_{len(list(tmp_path.iterdir()))}.s"
    inc = ["-I", os.path.join(REPO, "include"), "-I", B.CSRC]
    # (a quoted #include is looked up beside the including file first: a doctored header needs the unit beside it)
    src = os.path.join(str(src_dir) if src_dir else B.CSRC, "mctq_affine.hip")
    subprocess.run([hipcc, *B.FLAGS, *inc, "--cuda-device-only", "-S", "-o", str(asm), src], check=True, capture_output=True)
    return asm.read_text()


def check_headline_schedule(text):
    """What config 2's 0.76 of the HBM roofline rests on, asserted on the ISA of rows_kernel<AffineOp, float, float, 4, 1>:
    the tile's four 16-byte non-temporal loads are issued before anything of the parameter fetch (the channel's scale / zero
    point by scalar loads, the reciprocal); per element exactly v_mul (x * inv), v_rndne, v_med3 and one v_fma (q * s + 0) --
    no per-element division, ONE IEEE reciprocal per tile; four non-temporal 16-byte stores; no LDS, no scratch, at most 64
    VGPRs (8 waves per SIMD)."""
    assert HEADLINE + ":" in text, "rows_kernel<AffineOp, float, float, 4, 1> is not instantiated"
    body = text[text.index(HEADLINE + ":"):]
    meta = body[body.index("s_endpgm"):]
    body = body[:body.index("s_endpgm")]
    ins = [ln.split(";")[0].strip() for ln in body.splitlines()]
    ins = [ln for ln in ins if ln and not ln.startswith(".") and not ln.endswith(":")]
    loads = [i for i, x in enumerate(ins) if x.startswith("global_load_dwordx4")]
    stores = [i for i, x in enumerate(ins) if x.startswith("global_store_dwordx4")]
    assert len(loads) >= 4 and len(stores) >= 4
    # (1) the hot tile's four loads: non-temporal, back to back as far as the parameter fetch is concerned
    head = ins[:loads[3] + 1]
    assert all(ins[i].rstrip().endswith(" nt") for i in loads[:4]), "the tile's loads are not non-temporal"

    def is_fetch(x):          # a table element by a ONE-dword scalar load (kernel arguments arrive as x2 / x4 / x8), a reciprocal, a division
        return bool(re.match(r"s_load_dword s\d+,", x)) or x.startswith(("v_rcp_f32", "v_div_"))
    first_load = loads[0]
    kernarg_dwords = [x for x in ins[:first_load] if re.match(r"s_load_dword s\d+,", x)]
    fetch_before = [x for x in head[first_load:] if is_fetch(x)] + [x for x in ins[:first_load] if x.startswith(("v_rcp_f32", "v_div_"))]
    assert not fetch_before, f"parameter fetch scheduled in front of the tile's fourth data load: {fetch_before}"
    assert len(kernarg_dwords) <= 1, kernarg_dwords            # (one scalar kernel argument may arrive by a dword load)
    # (2) the hot path's arithmetic: from the fourth load to the fourth store
    hot = ins[loads[3] + 1:stores[3] + 1]
    count = lambda pat: sum(bool(re.match(pat, x)) for x in hot)
    assert count(r"v_rndne_f32") == 16 and count(r"v_med3_f32") == 16, (count(r"v_rndne_f32"), count(r"v_med3_f32"))
    assert count(r"v_fma_f32 v\d+, v\d+, s\d+, 0$") == 16, "one v_fma (q * s + 0) per element"
    # (x * inv per element; one more v_mul belongs to the tile's IEEE reciprocal; 0x4f7ffffe marks the row-index division)
    assert sum(x.startswith("v_mul_f32") and "0x4f7ffffe" not in x for x in hot) == 16 + 1, "one v_mul (x * inv) per element"
    assert count(r"v_div_fixup_f32") == 1 and count(r"v_rcp_f32") == 1, "ONE reciprocal per tile, none per element"
    assert sum(x.startswith("global_store_dwordx4") and x.rstrip().endswith(" nt") for x in hot) == 4, "non-temporal stores"
    assert any(is_fetch(x) for x in hot), "the parameter fetch runs under the loads' latency"
    # (3) resources
    assert not [x for x in ins if x.startswith(("ds_", "scratch_", "buffer_"))]
    m = re.search(re.escape(HEADLINE) + r"\.num_vgpr, (\d+)", meta)
    assert m and int(m.group(1)) <= 64, m and m.group(1)


def check_paced_schedule(text):
    """flat_paced_kernel (per-tensor launches of 3/4 ... 1 round) is nothing but an order of waits; on the ISA of the bfloat16 and the
    float32 instance: the tile's four non-temporal loads with an s_sleep between them, ONE s_waitcnt vmcnt(0) between the fourth load
    and the first rounding, an s_waitcnt vmcnt(0) between every two stores of the tile, at most 64 VGPRs."""
    for name in ("_ZN4mctq17flat_paced_kernelIDF16bLi2EEEvPKT_PS1_lNS_8AffineOpENS5_5ParamE",
                 "_ZN4mctq17flat_paced_kernelIfLi2EEEvPKT_PS1_lNS_8AffineOpENS5_5ParamE"):
        assert name + ":" in text, f"{name} is not instantiated"
        body = text[text.index(name + ":"):]
        body = body[:body.index(".Lfunc_end")]
        ins = [ln.split(";")[0].strip() for ln in body.splitlines()]
        ins = [ln for ln in ins if ln and not ln.startswith(".") and not ln.endswith(":")]
        loads = [i for i, x in enumerate(ins) if x.startswith("global_load_dwordx4")]
        stores = [i for i, x in enumerate(ins) if x.startswith("global_store_dwordx4")]
        assert len(loads) >= 4 and len(stores) >= 4
        for a, b in zip(loads[:3], loads[1:4]):
            assert any(x.startswith("s_sleep") for x in ins[a:b]), "the tile's loads are not spaced"
        first_round = next(i for i, x in enumerate(ins) if x.startswith("v_rndne_f32"))
        assert loads[3] < first_round and any(x.startswith("s_waitcnt vmcnt(0)") for x in ins[loads[3]:first_round]), "arithmetic before all loads landed"
        for a, b in zip(stores[:3], stores[1:4]):
            assert any(x.startswith("s_waitcnt vmcnt(0)") for x in ins[a:b]), "two stores without a completed one between them"
        m = re.search(re.escape(name) + r"\.num_vgpr, (\d+)", text)
        assert m and int(m.group(1)) <= 64, m and m.group(1)


def test_headline_kernel_keeps_the_schedule_its_roofline_figure_rests_on(tmp_path):
    """... and the check has teeth: the same translation unit compiled against a copy of the kernel header in which the
    parameter fetch is written IN FRONT of the tile's loads fails it."""
    from mct_quantizers_amd.hip import build as B
    text = _compile_affine_asm(tmp_path)
    check_headline_schedule(text)
    check_paced_schedule(text)
    hdr = open(os.path.join(B.CSRC, "mctq_kernels.hpp")).read()
    a = "  typename Op::Book book;\n  if constexpr (HasPrefetch<Op>::value) {"
    b = "  const typename Op::Param p = get_param();\n  const bool fast = __builtin_amdgcn_readfirstlane"
    assert hdr.count(a) == 1 and hdr.count(b) == 1
    bad = hdr.replace(a, "  const typename Op::Param p = get_param();\n  __builtin_amdgcn_sched_barrier(0);\n" + a)
    bad = bad.replace(b, "  const bool fast = __builtin_amdgcn_readfirstlane")
    d = tmp_path / "reordered"
    d.mkdir()
    (d / "mctq_kernels.hpp").write_text(bad)
    (d / "mctq_affine.hip").write_text(open(os.path.join(B.CSRC, "mctq_affine.hip")).read())
    with pytest.raises(AssertionError, match="in front of the tile's fourth data load"):
        check_headline_schedule(_compile_affine_asm(tmp_path, src_dir=d))
