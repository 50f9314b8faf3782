"""float64 tensors through the DOUBLE threshold list: the reference's LUT chain runs quotient, clip and
distances in double for a double tensor (quantizer_utils.py:126-134 under type promotion); the list built on the host by
bisecting the literal double scan must reproduce that scan for every t, and the kernel must equal the literal float64
kernel and the reference's float64 fixtures."""
import os
import struct

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_json

BOOKS = [
    ([-128.0, -96.0, -64.0, -40.0, -24.0, -12.0, -5.0, 0.0, 5.0, 12.0, 24.0, 40.0, 64.0, 96.0, 120.0, 127.0], 8, True),
    ([3.0, -7.0, 3.0, 100.0, -100.0, 0.0, 1.0], 8, True),                       # duplicates, list order != sorted order
    ([0.0, 17.0, 200.0, 255.0], 8, False),
    (list(range(-2000, 2001, 125)), 12, True),
    ([-30000.0, -1.0, 0.0, 2.0, 5.0, 31111.0], 16, True),
]


def _domain(bits, signed):
    return float(2 ** (bits - int(signed))), (float(-2 ** (bits - 1)) if signed else 0.0), float(2 ** (bits - 1) - 1 if signed else 2 ** bits - 1)


def _literal64(t, lut):
    d = np.abs(t[:, None] - np.asarray(lut, np.float64)[None, :])
    return np.asarray(lut, np.float64)[np.argmin(d, axis=1)]                   # numpy argmin: first minimum, as torch


def _parse(blob, P):
    T = np.frombuffer(blob[: 8 * P].tobytes(), dtype=np.float64)
    Q = np.frombuffer(blob[8 * P: 12 * P].tobytes(), dtype=np.float32)
    q_nan, p = struct.unpack("<2f", blob[12 * P: 12 * P + 8].tobytes())
    assert int(p) == P
    return T, Q, q_nan


def test_double_threshold_list_reproduces_the_literal_double_scan():
    from mct_quantizers_amd.hip import native
    rng = np.random.default_rng(5)
    for lut, bits, signed in BOOKS:
        mult, cmin, cmax = _domain(bits, signed)
        built = native.build_lut_steps_f64(lut, mult, cmin, cmax)
        assert built is not None, lut
        blob, P = built
        T, Q, q_nan = _parse(blob, P)
        assert q_nan == np.float32(lut[0]) / np.float32(mult)
        # probes: every threshold and its neighbours, exact midpoints of adjacent centres, random points, clip edges
        pts = [cmin, cmax]
        for t in T[1:]:
            if np.isfinite(t):
                pts += [np.nextafter(t, -np.inf), t, np.nextafter(t, np.inf)]
        vs = sorted(set(lut))
        pts += [(a + b) / 2 for a, b in zip(vs, vs[1:])]
        pts += list(rng.uniform(cmin, cmax, 20000))
        t = np.clip(np.asarray(pts, np.float64), cmin, cmax)
        idx = np.zeros(t.size, np.int64)
        s = P >> 1
        while s:
            idx += np.where(t >= T[idx + s], s, 0)
            s >>= 1
        got = Q[idx].astype(np.float64) * mult
        want = _literal64(t, lut)
        bad = np.flatnonzero(got != want)
        assert bad.size == 0, (lut[:4], t[bad[:3]], got[bad[:3]], want[bad[:3]])
    # a non-integer codebook does not qualify (the literal kernel stays)
    assert native.build_lut_steps_f64([0.5, 1.0, 7.25], 128.0, -128.0, 127.0) is None


@pytest.mark.gpu
def test_float64_list_kernel_equals_the_literal_double_kernel_and_the_oracle():
    from mct_quantizers_amd.hip import native
    from oracle import mctq_oracle as O
    lib = native.load()
    st = torch.cuda.current_stream().cuda_stream
    rng = np.random.default_rng(9)
    for lut, bits, signed in BOOKS:
        mult, cmin, cmax = _domain(bits, signed)
        blob, P = native.build_lut_steps_f64(lut, mult, cmin, cmax)
        T, _, _ = _parse(blob, P)
        dblob = torch.from_numpy(blob).cuda()
        dlut = torch.tensor(lut, dtype=torch.float32, device="cuda")
        for shape, axis in (((37, 1031), 0), ((5, 64, 9), 1), ((4099,), None), ((3, 2, 2, 6), 3)):
            n = int(np.prod(shape))
            thr = rng.uniform(0.5, 3.0, size=1 if axis is None else shape[axis]).astype(np.float32)
            x = rng.standard_normal(shape) * 2.0
            # plant values that land exactly on and next to thresholds of the first channel
            fin = T[1:][np.isfinite(T[1:])]
            if fin.size:
                d0 = np.float64(np.float32(thr[0] + np.float32(1e-8)))
                seeds = np.concatenate([fin, np.nextafter(fin, np.inf), np.nextafter(fin, -np.inf)]) / mult * d0
                flat = x.reshape(-1)
                flat[: min(seeds.size, flat.size)] = seeds[: flat.size]
            x[tuple(0 for _ in shape)] = np.nan
            xd = torch.from_numpy(np.ascontiguousarray(x)).cuda()
            y_list = torch.empty(shape, dtype=torch.float32, device="cuda")
            y_lit = torch.empty_like(y_list)
            if axis is None:
                d = float(np.float32(thr[0]) + np.float32(1e-8))
                assert lib.mctq_luts_per_tensor_f64(xd.data_ptr(), y_list.data_ptr(), n, d, float(thr[0]), dblob.data_ptr(), P,
                                                    mult, cmin, cmax, st) == 0, lib.mctq_last_error()
                assert lib.mctq_lut_per_tensor_f64(xd.data_ptr(), y_lit.data_ptr(), n, d, float(thr[0]), dlut.data_ptr(), len(lut),
                                                   mult, cmin, cmax, st) == 0
            else:
                outer, c, inner = int(np.prod(shape[:axis])), shape[axis], int(np.prod(shape[axis + 1:]))
                dthr = torch.from_numpy(thr).cuda()
                assert lib.mctq_luts_per_channel_f64(xd.data_ptr(), y_list.data_ptr(), outer, c, inner, dthr.data_ptr(), 1e-8,
                                                     dblob.data_ptr(), P, mult, cmin, cmax, st) == 0, lib.mctq_last_error()
                assert lib.mctq_lut_per_channel(xd.data_ptr(), y_lit.data_ptr(), outer, c, inner, native.DT_F64, dthr.data_ptr(),
                                                1e-8, dlut.data_ptr(), len(lut), mult, cmin, cmax, st) == 0
            torch.cuda.synchronize()
            a, b = y_list.cpu().numpy(), y_lit.cpu().numpy()
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (lut[:3], shape, axis)
            if axis is not None:
                want = O.lut_quantize_f64(x, lut, thr, signed, bits, 1e-8, per_channel=True, channel_axis=axis)
                m = np.isfinite(x)
                assert np.array_equal(a.view(np.uint32)[m], np.asarray(want, np.float32).view(np.uint32)[m])
    assert "lut64_steps_kernel" in native.last_launch()


@pytest.mark.gpu
def test_float64_reference_fixtures_take_the_list_kernel():
    """The 41 float64 cases produced by the reference: the LUT ones now run the double threshold list."""
    import warnings
    import mct_quantizers_amd as mq
    from mct_quantizers_amd.hip import native
    meta = load_json("cases_f64.json")
    arrays = np.load(os.path.join(GOLDEN, "cases_f64.npz"))
    seen = 0
    for c in meta["cases"]:
        if "LUT" not in c["cls"] and "Lut" not in c["cls"]:
            continue
        x_np, want = arrays[c["id"] + "_x"], arrays[c["id"] + "_y"]
        x = torch.from_numpy(np.ascontiguousarray(x_np)).cuda()
        if c["memory_format"] == "channels_last":
            x = x.contiguous(memory_format=torch.channels_last)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            q = getattr(mq.pytorch_quantizers, c["cls"])(**c["kwargs"])
        got = q(x).cpu().numpy()
        assert got.dtype == want.dtype and np.array_equal(got.view(np.uint32), want.view(np.uint32)), c["id"]
        seen += "lut64_steps_kernel" in native.last_launch()
    assert seen >= 5
