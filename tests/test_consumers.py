"""Integer consumer of the codes (mctq_qlinear_i8 / consumers.QuantizedLinear).

Oracle: exact int64 product of the codes, scaled once in float32 (oracle/mctq_oracle.py::qlinear_i8).  Bar: the
kernel is bit-identical to it for every shape / tail / code type / launch variant; against the reference-style
float32 path (fake-quant both operands, F.linear) it agrees to the float32 rounding of that path's long sum.
"""
import warnings

import numpy as np
import pytest
import torch

from conftest import bits_equal, first_mismatch


def _problem(rng, M, N, K, u8, with_bias=True):
    a = rng.integers(0, 256, (M, K)).astype(np.uint8) if u8 else rng.integers(-128, 128, (M, K)).astype(np.int8)
    w = rng.integers(-128, 128, (N, K)).astype(np.int8)
    za = int(rng.integers(0, 256)) if u8 else int(rng.integers(-128, 128))
    sa = float(rng.uniform(0.001, 0.1))
    ws = rng.uniform(0.001, 0.1, N).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32) if with_bias else None
    return a, za, sa, w, ws, bias


def _model(K=64, N=24, bits_w=8, bits_a=8, per_channel=True, act="uniform", seed=0):
    import mct_quantizers_amd as mq
    Q = mq.pytorch_quantizers
    torch.manual_seed(seed)
    lin = torch.nn.Linear(K, N)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        if per_channel:
            thr = [float(v) for v in lin.weight.detach().abs().max(1).values]
            wq = Q.WeightsSymmetricInferableQuantizer(num_bits=bits_w, threshold=thr, per_channel=True, channel_axis=0)
        else:
            wq = Q.WeightsPOTInferableQuantizer(num_bits=bits_w, threshold=[0.25], per_channel=False)
        if act == "uniform":
            aq = Q.ActivationUniformInferableQuantizer(num_bits=bits_a, min_range=[-2.5], max_range=[3.1])
        elif act == "signed":
            aq = Q.ActivationSymmetricInferableQuantizer(num_bits=bits_a, threshold=[3.3], signed=True)
        else:
            aq = Q.ActivationPOTInferableQuantizer(num_bits=bits_a, threshold=[4.0], signed=False)
    return torch.nn.Sequential(mq.PytorchActivationQuantizationHolder(aq),
                               mq.PytorchQuantizationWrapper(lin, {"weight": wq}))


# ------------------------------------------------------------------------------------------------
# CPU
# ------------------------------------------------------------------------------------------------

def test_oracle_large_case_branch_equals_the_integer_branch():
    from oracle import mctq_oracle as O
    rng = np.random.default_rng(9)
    a, za, sa, w, ws, bias = _problem(rng, 64, 96, 4096, True)                # 2^24.6 products: float64-BLAS branch
    got = O.qlinear_i8(a, za, sa, w, ws, bias)
    acc = (a.astype(np.int64) - za) @ w.astype(np.int64).T
    want = (acc.astype(np.int32).astype(np.float32) * (np.float32(sa) * ws)).astype(np.float32) + bias
    assert bits_equal(got, want.astype(np.float32))
    a[:] = 255
    w[:] = -128
    got = O.qlinear_i8(a, 0, sa, w, ws, None)
    assert np.all(got == (np.float32(-255 * 128 * 4096) * (np.float32(sa) * ws)).astype(np.float32)[None, :])


def test_oracle_is_the_exact_product_of_the_dequantized_operands():
    from oracle import mctq_oracle as O
    rng = np.random.default_rng(3)
    for u8 in (False, True):
        a, za, sa, w, ws, bias = _problem(rng, 9, 20, 96, u8)
        y = O.qlinear_i8(a, za, sa, w, ws, bias)
        exact = ((a.astype(np.float64) - za) * np.float32(sa).astype(np.float64)) @ (w.astype(np.float64) * ws.astype(np.float64)[:, None]).T + bias
        assert np.allclose(y, exact, rtol=3e-7, atol=1e-6 * np.abs(exact).max())


@pytest.mark.parametrize("act", ["uniform", "signed", "unsigned_pot"])
@pytest.mark.parametrize("per_channel", [True, False])
def test_quantized_linear_on_cpu_matches_oracle_and_reference_path(act, per_channel):
    from oracle import mctq_oracle as O
    from mct_quantizers_amd import consumers
    from mct_quantizers_amd.hip import ops
    model = _model(act=act, per_channel=per_channel)
    x = torch.randn(3, 5, 64) * 1.5
    ref = model(x)                                                  # fake-quant + float32 F.linear
    assert consumers.fuse_linear_consumers(model) == 1
    ql = model[1]
    assert isinstance(model[0], torch.nn.Identity) and isinstance(ql, consumers.QuantizedLinear)
    y = model(x)
    assert y.shape == ref.shape == (3, 5, 24)
    assert torch.allclose(y, ref, rtol=1e-5, atol=2e-6 * float(ref.detach().abs().max()))
    a_codes = ops.fq_codes(x.reshape(-1, 64), None, None, None, ql._a_qmin, ql._a_qmax, ql._a_scale, ql._a_zp)
    want = O.qlinear_i8(a_codes.numpy(), ql._a_zp, ql._a_scale, ql._w_codes.numpy(), ql._w_scales.numpy(),
                        ql.bias.detach().numpy())
    assert bits_equal(y.detach().reshape(-1, 24).numpy(), want)


def test_weight_codes_follow_in_place_weight_updates():
    from mct_quantizers_amd import consumers
    model = _model()
    consumers.fuse_linear_consumers(model)
    ql = model[1]
    x = torch.randn(4, 64)
    y0 = model(x)
    codes0 = ql._w_codes.clone()
    with torch.no_grad():
        ql.weight.mul_(0.5)
    y1 = model(x)
    assert not torch.equal(codes0, ql._w_codes) and not torch.equal(y0, y1)


def test_fuse_leaves_unsupported_pairs_alone():
    import mct_quantizers_amd as mq
    from mct_quantizers_amd import consumers
    Q = mq.pytorch_quantizers
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        aq = Q.ActivationSymmetricInferableQuantizer(num_bits=8, threshold=[2.0], signed=True)
        conv = mq.PytorchQuantizationWrapper(torch.nn.Conv2d(3, 4, 3), {"weight": Q.WeightsSymmetricInferableQuantizer(
            num_bits=8, threshold=[1.0], per_channel=False)})
        odd = mq.PytorchQuantizationWrapper(torch.nn.Linear(24, 8), {"weight": Q.WeightsSymmetricInferableQuantizer(
            num_bits=8, threshold=[1.0], per_channel=False)})                    # K % 16 != 0
        uni = mq.PytorchQuantizationWrapper(torch.nn.Linear(32, 8), {"weight": Q.WeightsUniformInferableQuantizer(
            num_bits=8, min_range=[-1.0], max_range=[1.0], per_channel=False)})   # zero point != 0
    model = torch.nn.Sequential(mq.PytorchActivationQuantizationHolder(aq), conv,
                                torch.nn.Sequential(mq.PytorchActivationQuantizationHolder(aq), odd),
                                torch.nn.Sequential(mq.PytorchActivationQuantizationHolder(aq), uni))
    assert consumers.fuse_linear_consumers(model) == 0


# ------------------------------------------------------------------------------------------------
# GPU
# ------------------------------------------------------------------------------------------------

def _run_kernel(lib, native, a, u8, za, sa, w, ws, bias):
    from oracle import mctq_oracle as O  # noqa: F401
    dev = torch.device("cuda")
    M, K = a.shape
    N = w.shape[0]
    at, wt, wst = (torch.from_numpy(v).to(dev) for v in (a, w, ws))
    rs = wt.sum(dim=1, dtype=torch.int32)
    bt = None if bias is None else torch.from_numpy(bias).to(dev)
    y = torch.full((M, N), float("nan"), dtype=torch.float32, device=dev)
    rc = lib.mctq_qlinear_i8(at.data_ptr(), native.CODE_U8 if u8 else native.CODE_I8, za, sa, wt.data_ptr(), wst.data_ptr(),
                             rs.data_ptr(), None if bt is None else bt.data_ptr(), y.data_ptr(), M, N, K,
                             torch.cuda.current_stream().cuda_stream)
    assert rc == 0, lib.mctq_last_error()
    return y.cpu().numpy()


@pytest.mark.gpu
@pytest.mark.parametrize("variant", [0, 181, 182, 184, 83233, 86433, 86633, 812613, 166623, 1612623, 612, 1212, 662])      # every
# kernel the automatic choice can select for ragged shapes (ABI v8: only those are built; the whole-tile kernels 2544 / 2548 /
# 2560 have their own test below)
def test_qlinear_kernel_is_bit_exact_against_the_integer_oracle(variant):
    from oracle import mctq_oracle as O
    from mct_quantizers_amd.hip import native
    lib = native.load()
    rng = np.random.default_rng(100 + variant)
    assert lib.mctq_set_tuning(b"ql_variant", variant) == 0
    try:
        shapes = [(1, 16, 16), (5, 100, 256), (16, 33, 272), (17, 16, 4096), (33, 64, 2048), (64, 4096, 1024),
                  (100, 48, 11008), (7, 1000, 4112), (130, 20, 528), (2, 3, 32768), (129, 130, 144), (300, 257, 1040),
                  (512, 256, 4096)]
        for (M, N, K) in shapes:
            for u8 in (False, True):
                a, za, sa, w, ws, bias = _problem(rng, M, N, K, u8, with_bias=(M + N) % 2 == 1)
                if K == 32768:                                  # extreme codes: the accumulator's worst case
                    a[:] = 255 if u8 else -128
                    w[:] = -128
                    za = 0 if u8 else 127
                got = _run_kernel(lib, native, a, u8, za, sa, w, ws, bias)
                want = O.qlinear_i8(a, za, sa, w, ws, bias)
                assert bits_equal(got, want), f"variant {variant} M={M} N={N} K={K} u8={u8}: {first_mismatch(got, want)}"
    finally:
        lib.mctq_set_tuning(b"ql_variant", 0)


@pytest.mark.gpu
def test_qlinear_automatic_kernel_choice_is_bit_exact_across_the_cost_models_regimes():
    """The dispatcher picks among the streaming kernels, the 8- / 16-wave ring tiles, the two-buffer tiles and the
    asm-pinned kernels by a cost model (csrc/mctq_qlinear.hip: qlinear_dispatch): shapes that land on every candidate, with
    tails in M / N / K, both code types -- each equal to the integer oracle bit for bit, and every candidate reached."""
    from oracle import mctq_oracle as O
    from mct_quantizers_amd.hip import native
    lib = native.load()
    assert lib.mctq_set_tuning(b"ql_variant", 0) == 0
    rng = np.random.default_rng(4242)
    shapes = [(8, 4096, 2048), (32, 4096, 2048), (16, 11008, 2048), (64, 4096, 2048), (96, 4096, 2064), (128, 4096, 4096),
              (200, 4096, 2048), (250, 4090, 2048), (320, 4096, 2048), (448, 4096, 4096), (640, 4096, 2048), (700, 4096, 1040),
              (1100, 4096, 2048), (1900, 4096, 1024), (1024, 4096, 1024), (2048, 4096, 512), (4096, 4096, 256)]
    seen = set()
    for (M, N, K) in shapes:
        u8 = (M + K) % 3 != 0
        a, za, sa, w, ws, bias = _problem(rng, M, N, K, u8, with_bias=M % 2 == 0)
        got = _run_kernel(lib, native, a, u8, za, sa, w, ws, bias)
        seen.add(native.last_launch().split("<")[0])
        want = O.qlinear_i8(a, za, sa, w, ws, bias)
        assert bits_equal(got, want), f"M={M} N={N} K={K} u8={u8} [{native.last_launch()}]: {first_mismatch(got, want)}"
    assert len(seen) >= 8, sorted(seen)             # streaming, ring tiles of several shapes and wave counts, two-buffer, wide


@pytest.mark.gpu
def test_qlinear_rejects_bad_arguments():
    from mct_quantizers_amd.hip import native
    lib = native.load()
    dev = torch.device("cuda")
    a = torch.zeros(4, 64, dtype=torch.int8, device=dev)
    w = torch.zeros(8, 64, dtype=torch.int8, device=dev)
    s = torch.ones(8, device=dev)
    r = torch.zeros(8, dtype=torch.int32, device=dev)
    y = torch.zeros(4, 8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    E = native.MCTQ_E_ARG
    call = lambda *args: lib.mctq_qlinear_i8(*args)   # noqa: E731
    assert call(a.data_ptr(), native.CODE_I8, 0, 1.0, w.data_ptr(), s.data_ptr(), r.data_ptr(), None, y.data_ptr(), 4, 8, 24, st) == E
    assert b"multiple of 16" in lib.mctq_last_error()
    assert call(a.data_ptr(), 77, 0, 1.0, w.data_ptr(), s.data_ptr(), r.data_ptr(), None, y.data_ptr(), 4, 8, 64, st) == E
    assert call(a.data_ptr() + 1, native.CODE_I8, 0, 1.0, w.data_ptr(), s.data_ptr(), r.data_ptr(), None, y.data_ptr(), 4, 8, 64, st) == E
    assert call(a.data_ptr(), native.CODE_I8, 0, 1.0, w.data_ptr(), s.data_ptr(), r.data_ptr(), None, y.data_ptr(), 4, 8, 65536, st) == E
    assert call(a.data_ptr(), native.CODE_I8, 0, 1.0, None, s.data_ptr(), r.data_ptr(), None, y.data_ptr(), 4, 8, 64, st) == E
    assert call(None, native.CODE_I8, 0, 1.0, None, None, None, None, None, 0, 8, 64, st) == 0        # empty


@pytest.mark.gpu
@pytest.mark.parametrize("act", ["uniform", "signed", "unsigned_pot"])
@pytest.mark.parametrize("K,N,batch", [(64, 24, (3, 5)), (4096, 512, (64,)), (1024, 1000, (1,))])
def test_quantized_linear_on_gpu(act, K, N, batch):
    from oracle import mctq_oracle as O
    from mct_quantizers_amd import consumers
    model = _model(K=K, N=N, act=act).cuda()
    x = (torch.randn(*batch, K) * 1.5).cuda()
    ref = model(x)                                                  # HIP fake-quant kernels + float32 GEMM
    assert consumers.fuse_linear_consumers(model) == 1
    ql = model[1]
    y = model(x)
    assert y.is_cuda and y.shape == ref.shape
    scale = float(ref.abs().max())
    assert torch.allclose(y, ref, rtol=1e-4, atol=3e-6 * scale * (K / 64) ** 0.5), float((y - ref).abs().max())
    # and bit-exact against the oracle on the codes the module produced
    from mct_quantizers_amd.hip import ops
    a_codes = ops.fq_codes(x.reshape(-1, K), None, None, None, ql._a_qmin, ql._a_qmax, ql._a_scale, ql._a_zp)
    want = O.qlinear_i8(a_codes.cpu().numpy(), ql._a_zp, ql._a_scale, ql._w_codes.cpu().numpy(),
                        ql._w_scales.cpu().numpy(), ql.bias.detach().cpu().numpy())
    got = y.detach().reshape(-1, N).cpu().numpy()
    assert bits_equal(got, want), first_mismatch(got, want)


@pytest.mark.gpu
def test_quantized_linear_replays_in_a_hip_graph():
    from mct_quantizers_amd import consumers
    model = _model(K=1024, N=256).cuda()
    consumers.fuse_linear_consumers(model)
    x = torch.randn(16, 1024, device="cuda")
    want = model(x)
    static_x = x.clone()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        model(static_x)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = model(static_x)
    static_x.copy_(x * 0.5)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, model(x * 0.5)) and not torch.equal(out, want)


def _stack(layers=3, d=64, seed=1):
    import mct_quantizers_amd as mq
    Q = mq.pytorch_quantizers
    torch.manual_seed(seed)
    mods = []
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for i in range(layers):
            lin = torch.nn.Linear(d, d)
            thr = [float(v) for v in lin.weight.detach().abs().max(1).values]
            aq = (Q.ActivationUniformInferableQuantizer(num_bits=8, min_range=[-2.0], max_range=[2.5]) if i % 2 == 0
                  else Q.ActivationSymmetricInferableQuantizer(num_bits=8, threshold=[2.0], signed=True))
            mods += [mq.PytorchActivationQuantizationHolder(aq),
                     mq.PytorchQuantizationWrapper(lin, {"weight": Q.WeightsSymmetricInferableQuantizer(
                         num_bits=8, threshold=thr, per_channel=True, channel_axis=0)})]
    return torch.nn.Sequential(*mods)


def _check_chain(device):
    from mct_quantizers_amd import consumers
    plain, chained = _stack().to(device), _stack().to(device)
    assert consumers.fuse_linear_consumers(plain) == 3
    assert consumers.fuse_linear_consumers(chained, chain=True) == 3
    assert chained[1].emit_codes_for is not None and chained[3].emit_codes_for is not None
    assert chained[5].emit_codes_for is None                       # last layer: float32 out
    x = (torch.randn(7, 64) * 1.5).to(device)
    y0, y1 = plain(x), chained(x)
    assert y1.dtype == torch.float32 and torch.equal(y0, y1)       # same codes in between, bit for bit
    mid = chained[:2](x)
    assert mid.dtype == torch.int8                                  # layer 1 feeds a signed 8-bit quantizer


def test_chained_consumers_pass_codes_between_layers_cpu():
    _check_chain("cpu")


@pytest.mark.gpu
def test_chained_consumers_pass_codes_between_layers_gpu():
    _check_chain("cuda")


@pytest.mark.gpu
@pytest.mark.parametrize("variant", [2560, 2548, 2544, 0])
def test_dense_tile_kernels_are_bit_exact_and_requantize(variant):
    """Whole-tile shapes of the many-rows kernels (256 x 256 ping-pong, wave-wide 256 x 256 / 128 x 256 / 256 x 128):
    exact against the int64 oracle for int8 and uint8 activation codes (extreme codes at the longest K included), the
    requantizing epilogue equal to the codes kernel, shapes they cannot take refused; variant 0 = what the dispatch picks
    for a chip-filling shape."""
    from oracle import mctq_oracle as O
    from mct_quantizers_amd import consumers
    from mct_quantizers_amd.hip import native, ops
    lib = native.load()
    rng = np.random.default_rng(variant + 7)
    dev = torch.device("cuda")
    assert lib.mctq_set_tuning(b"ql_variant", variant) == 0
    try:
        shapes = [(256, 256, 256), (256, 512, 384), (768, 512, 640), (512, 256, 32768), (1024, 768, 1152)]
        if variant == 0:
            shapes = [(4096, 4096, 256), (2048, 4096, 384)]
        for (M, N, K) in shapes:
            for u8 in (False, True):
                a, za, sa, w, ws, bias = _problem(rng, M, N, K, u8, with_bias=(M // 256 + N // 256) % 2 == 1)
                if K == 32768:
                    a[:] = 255 if u8 else -128
                    w[:] = -128
                    za = 0 if u8 else 127
                got = _run_kernel(lib, native, a, u8, za, sa, w, ws, bias)
                want = O.qlinear_i8(a, za, sa, w, ws, bias)
                assert bits_equal(got, want), f"variant {variant} M={M} N={N} K={K} u8={u8}: {first_mismatch(got, want)}"
        # requantizing epilogue (int8 / uint8 / 4-bit-range codes of the next layer)
        M, N, K = 512, 512, 512
        a, za, sa, w, ws, bias = _problem(rng, M, N, K, True)
        at, wt, wst, bt = (torch.from_numpy(v).to(dev) for v in (a, w, ws, bias))
        rs = wt.sum(dim=1, dtype=torch.int32)
        y = consumers.qlinear_i8(at, za, sa, wt, wst, rs, bt)
        for out in ((0.05, 3, -128, 127), (0.11, 100, 0, 255), (0.5, 0, -8, 7)):
            got = consumers.qlinear_i8(at, za, sa, wt, wst, rs, bt, out)
            want_codes = ops.fq_codes(y, None, None, None, out[2], out[3], out[0], out[1])
            assert got.dtype == want_codes.dtype and torch.equal(got, want_codes), out
        if variant:
            a, za, sa, w, ws, bias = _problem(rng, 300, 256, 256, True)                 # ragged rows: refused, not misread
            at, wt, wst = (torch.from_numpy(v).to(dev) for v in (a, w, ws))
            rs = wt.sum(dim=1, dtype=torch.int32)
            yb = torch.empty(300, 256, device=dev)
            rc = lib.mctq_qlinear_i8(at.data_ptr(), native.CODE_U8, za, sa, wt.data_ptr(), wst.data_ptr(), rs.data_ptr(), None,
                                     yb.data_ptr(), 300, 256, 256, torch.cuda.current_stream().cuda_stream)
            assert rc == native.MCTQ_E_ARG
    finally:
        lib.mctq_set_tuning(b"ql_variant", 0)


@pytest.mark.gpu
@pytest.mark.parametrize("M", [5, 64, 300])
def test_requantizing_epilogue_equals_the_codes_kernel(M):
    from mct_quantizers_amd import consumers
    from mct_quantizers_amd.hip import ops
    rng = np.random.default_rng(M)
    N, K = 200, 512
    a, za, sa, w, ws, bias = _problem(rng, M, N, K, True)
    dev = torch.device("cuda")
    at, wt, wst, bt = (torch.from_numpy(v).to(dev) for v in (a, w, ws, bias))
    rs = wt.sum(dim=1, dtype=torch.int32)
    y = consumers.qlinear_i8(at, za, sa, wt, wst, rs, bt)
    for out in ((0.05, 3, -128, 127), (0.11, 100, 0, 255), (0.5, 0, -8, 7)):
        got = consumers.qlinear_i8(at, za, sa, wt, wst, rs, bt, out)
        want = ops.fq_codes(y, None, None, None, out[2], out[3], out[0], out[1])
        assert got.dtype == want.dtype and torch.equal(got, want), out


def test_w4_consumer_layout_round_trip():
    from mct_quantizers_amd import consumers
    codes = torch.randint(-8, 8, (5, 32), dtype=torch.int8)
    packed = consumers.pack_w4(codes)
    assert packed.shape == (5, 16) and packed.dtype == torch.uint8
    b = packed.reshape(5, 4, 4).to(torch.int16)                      # [row][group of 8 k][byte j]
    lo, hi = b & 0xF, (b >> 4) & 0xF
    back = torch.cat((lo, hi), dim=2)                                # k = j, then k = j + 4
    back = torch.where(back > 7, back - 16, back).to(torch.int8).reshape(5, 32)
    assert torch.equal(back, codes)


@pytest.mark.gpu
@pytest.mark.parametrize("M", [1, 16, 17, 33, 64, 130])
def test_w4a8_kernel_is_bit_exact_against_the_integer_oracle(M):
    from oracle import mctq_oracle as O
    from mct_quantizers_amd import consumers
    from mct_quantizers_amd.hip import ops
    rng = np.random.default_rng(M)
    dev = torch.device("cuda")
    for (N, K) in [(16, 16), (100, 256), (33, 272), (64, 4096), (1000, 4112), (48, 11008)]:
        for u8 in (False, True):
            a, za, sa, _, ws, bias = _problem(rng, M, N, K, u8, with_bias=(M + N) % 2 == 1)
            w = rng.integers(-8, 8, (N, K)).astype(np.int8)
            if K == 11008:
                w[:] = -8                                            # extreme codes
            at, wt, wst = (torch.from_numpy(v).to(dev) for v in (a, w, ws))
            bt = None if bias is None else torch.from_numpy(bias).to(dev)
            rs = wt.sum(dim=1, dtype=torch.int32)
            got = consumers.qlinear_w4a8(at, za, sa, consumers.pack_w4(wt), wst, rs, bt)
            want = O.qlinear_i8(a, za, sa, w, ws, bias)
            assert bits_equal(got.cpu().numpy(), want), f"M={M} N={N} K={K} u8={u8}: {first_mismatch(got.cpu().numpy(), want)}"
            out = (0.07, 5, -128, 127)
            codes = consumers.qlinear_w4a8(at, za, sa, consumers.pack_w4(wt), wst, rs, bt, out)
            assert torch.equal(codes, ops.fq_codes(got, None, None, None, out[2], out[3], out[0], out[1]))


@pytest.mark.gpu
def test_quantized_linear_streams_packed_4bit_weights_for_small_batches():
    from mct_quantizers_amd import consumers
    model = _model(K=1024, N=256, bits_w=4).cuda()
    ref_model = _model(K=1024, N=256, bits_w=4).cuda()
    assert consumers.fuse_linear_consumers(model) == 1
    ql = model[1]
    for batch in (8, 200):                                           # packed path, then the int8 tiled path
        x = torch.randn(batch, 1024, device="cuda") * 1.5
        y, ref = model(x), ref_model(x)
        assert ql._w_codes4 is not None and ql._w_codes4.shape == (256, 512)
        assert torch.allclose(y, ref, rtol=1e-4, atol=1e-5 * float(ref.detach().abs().max()))
    # both paths compute the same integers: identical results for the same rows
    x = torch.randn(32, 1024, device="cuda")
    y_small = model(x)
    y_big = model(torch.cat([x, x, x]))[:32]
    assert torch.equal(y_small, y_big)


class _Branchy(torch.nn.Module):
    """x -> holder -> wrapped Linear -> (+ x) -> holder -> {wrapped Linear, wrapped Linear}: the second holder has two
    consumers and must stay; the first pair fuses."""

    def __init__(self):
        super().__init__()
        import mct_quantizers_amd as mq
        Q = mq.pytorch_quantizers
        torch.manual_seed(5)

        def wrapped(n_in, n_out):
            lin = torch.nn.Linear(n_in, n_out)
            thr = [float(v) for v in lin.weight.detach().abs().max(1).values]
            return mq.PytorchQuantizationWrapper(lin, {"weight": Q.WeightsSymmetricInferableQuantizer(
                num_bits=8, threshold=thr, per_channel=True, channel_axis=0)})
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            self.h1 = mq.PytorchActivationQuantizationHolder(Q.ActivationUniformInferableQuantizer(8, [-2.0], [2.5]))
            self.l1 = wrapped(64, 64)
            self.h2 = mq.PytorchActivationQuantizationHolder(Q.ActivationSymmetricInferableQuantizer(8, [4.0], True))
            self.l2a, self.l2b = wrapped(64, 32), wrapped(64, 32)
            self.h3 = mq.PytorchActivationQuantizationHolder(Q.ActivationSymmetricInferableQuantizer(8, [4.0], True))
            self.l3 = wrapped(32, 16)

    def forward(self, x):
        y = self.l1(self.h1(x)) + x
        q = self.h2(y)
        z = self.l2a(q) * torch.sigmoid(self.l2b(q))
        return self.l3(self.h3(z))


def _check_fx_fusion(device):
    from mct_quantizers_amd import consumers
    model = _Branchy().to(device)
    x = (torch.randn(9, 64) * 1.5).to(device)
    ref = model(x)
    gm, n = consumers.fuse_linear_consumers_fx(model, chain=True)
    assert n == 2                                                  # (h1, l1) and (h3, l3); h2 feeds two layers
    kinds = [type(m).__name__ for m in gm.modules()]
    assert kinds.count("QuantizedLinear") == 2 and "PytorchActivationQuantizationHolder" in kinds
    y = gm(x)
    assert y.shape == ref.shape and torch.allclose(y, ref, rtol=1e-4, atol=1e-5 * float(ref.detach().abs().max()))


def test_fx_fusion_on_a_branching_model_cpu():
    _check_fx_fusion("cpu")


@pytest.mark.gpu
def test_fx_fusion_on_a_branching_model_gpu():
    _check_fx_fusion("cuda")


def _pointwise_block(c_in=32, c_mid=64, c_out=16, seed=3):
    """holder -> wrapped 1x1 conv -> holder -> wrapped 1x1 conv (the pointwise pair of an inverted-residual block)."""
    import mct_quantizers_amd as mq
    Q = mq.pytorch_quantizers
    torch.manual_seed(seed)
    mods = []
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for ci, co, aq in ((c_in, c_mid, Q.ActivationUniformInferableQuantizer(8, [-2.0], [2.5])),
                           (c_mid, c_out, Q.ActivationSymmetricInferableQuantizer(8, [6.0], False))):
            conv = torch.nn.Conv2d(ci, co, 1)
            thr = [float(v) for v in conv.weight.detach().abs().amax(dim=(1, 2, 3))]
            mods += [mq.PytorchActivationQuantizationHolder(aq),
                     mq.PytorchQuantizationWrapper(conv, {"weight": Q.WeightsSymmetricInferableQuantizer(
                         num_bits=8, threshold=thr, per_channel=True, channel_axis=0)})]
    return torch.nn.Sequential(*mods)


def _check_pointwise(device):
    from mct_quantizers_amd import consumers
    ref_model, model, chained = (_pointwise_block().to(device) for _ in range(3))
    assert consumers.fuse_linear_consumers(model) == 2 and consumers.fuse_linear_consumers(chained, chain=True) == 2
    assert isinstance(model[1], consumers.QuantizedConv1x1)
    for fmt in (torch.contiguous_format, torch.channels_last):
        x = (torch.randn(2, 32, 7, 5) * 1.5).to(device).contiguous(memory_format=fmt)
        ref, y, yc = ref_model(x), model(x), chained(x)
        assert y.shape == ref.shape == (2, 16, 7, 5)
        assert torch.allclose(y, ref, rtol=1e-4, atol=1e-5 * float(ref.detach().abs().max()))
        assert torch.equal(y, yc)                                    # codes handed from conv to conv: same result
    # layers the consumer cannot take stay as they are
    import mct_quantizers_amd as mq
    Q = mq.pytorch_quantizers
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        strided = torch.nn.Sequential(
            mq.PytorchActivationQuantizationHolder(Q.ActivationSymmetricInferableQuantizer(8, [2.0], True)),
            mq.PytorchQuantizationWrapper(torch.nn.Conv2d(32, 8, 1, stride=2), {"weight": Q.WeightsSymmetricInferableQuantizer(
                num_bits=8, threshold=[1.0], per_channel=False)}))
    assert consumers.fuse_linear_consumers(strided) == 0


def test_pointwise_convolution_consumer_cpu():
    _check_pointwise("cpu")


@pytest.mark.gpu
def test_pointwise_convolution_consumer_gpu():
    _check_pointwise("cuda")


def test_half_precision_layers_are_left_unfused_and_never_misread():
    """a half-precision wrapped Linear must not reach the integer consumer (it reads the bias as float32
    and answers in float32)."""
    import mct_quantizers_amd as mq
    from mct_quantizers_amd import consumers
    Q = mq.pytorch_quantizers
    lin = torch.nn.Linear(32, 8).half()
    wrapper = mq.PytorchQuantizationWrapper(lin, {"weight": Q.WeightsSymmetricInferableQuantizer(8, [0.5] * 8, True, 0)})
    holder = mq.PytorchActivationQuantizationHolder(Q.ActivationUniformInferableQuantizer(8, [-2.0], [2.0]))
    model = torch.nn.Sequential(holder, wrapper)
    assert consumers.fuse_linear_consumers(model) == 0 and isinstance(model[1], mq.PytorchQuantizationWrapper)
    with pytest.raises(TypeError):
        consumers.QuantizedLinear(torch.nn.Linear(32, 8).half(), Q.WeightsSymmetricInferableQuantizer(8, [0.5] * 8, True, 0),
                                  Q.ActivationUniformInferableQuantizer(8, [-2.0], [2.0]))
    with pytest.raises(TypeError):
        consumers._check_consumer_operands(torch.zeros(2, 16, dtype=torch.int8), torch.ones(8), torch.zeros(8, dtype=torch.int32),
                                           torch.zeros(8, dtype=torch.float16))
    # float32 layer whose bias was converted afterwards: converted back for the launch, same result
    lin32 = torch.nn.Linear(32, 8)
    ql = consumers.QuantizedLinear(lin32, Q.WeightsSymmetricInferableQuantizer(8, [0.5] * 8, True, 0),
                                   Q.ActivationUniformInferableQuantizer(8, [-2.0], [2.0]))
    x = torch.randn(3, 32)
    want = ql(x)
    ql.bias = torch.nn.Parameter(lin32.bias.detach().double())
    assert torch.equal(ql(x), want)                    # float32 -> float64 -> float32 is the identity on the values
