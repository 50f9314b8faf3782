"""rowsteps_kernel (csrc/mctq_kernels.hpp; tuning key "rowsteps": by default taken only where it measured faster -- a grid of
one round of blocks, profiles/r04/rowsteps_probe_sustained.log): per-channel rows of two or three whole 256-lane-vector steps -- float32 rows of 2048 / 3072
elements, 16-bit rows of 2048 / 4096 / 6144 -- four steps per block across row boundaries.  Against the oracle, for every
storage type, with outer > 1 (channel = row % channels), ragged last blocks, zero points, and equal to the rows_kernel."""
import warnings

import numpy as np
import pytest
import torch

import mct_quantizers_amd as mq
from mct_quantizers_amd.hip import native
from conftest import bits_equal, first_mismatch

pytestmark = pytest.mark.gpu
Q = mq.pytorch_quantizers
TORCH = {"float32": torch.float32, "float16": torch.float16, "bfloat16": torch.bfloat16}


def _oracle(cls, kw, x_np, dt):
    from oracle import oracle_call
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return oracle_call(cls, kw, x_np, in_dtype=dt)


@pytest.mark.parametrize("dt,shape,axis", [
    ("float32", (4096, 2048), 0), ("float32", (301, 2048), 0), ("float32", (77, 3072), 0), ("float32", (1, 2048), 0),
    ("float32", (3, 50, 2048), 1), ("float32", (2, 7, 64, 32), 1), ("float32", (5, 3072), 0),
    ("bfloat16", (512, 4096), 0), ("bfloat16", (33, 2048), 0), ("bfloat16", (10, 6144), 0), ("float16", (129, 4096), 0),
    ("float16", (3, 21, 2048), 1), ("bfloat16", (2, 3, 64, 64), 1),
])
@pytest.mark.parametrize("kind", ["symmetric", "uniform"])
def test_rowsteps_equals_oracle_and_rows_kernel(dt, shape, axis, kind):
    rng = np.random.default_rng(hash((dt, shape, axis, kind)) % (1 << 31))
    x32 = (rng.standard_normal(shape) * 1.5).astype(np.float32)
    x = torch.from_numpy(x32).to(TORCH[dt]).cuda()
    x_np = x.float().cpu().numpy()                        # the stored values, widened
    c = shape[axis]
    if kind == "symmetric":
        cls, kw = "WeightsSymmetricInferableQuantizer", dict(num_bits=8, threshold=[float(v) for v in rng.uniform(0.5, 4.0, c)],
                                                              per_channel=True, channel_axis=axis)
    else:
        lo = [float(v) for v in rng.uniform(-3.0, -0.5, c)]
        hi = [float(v) for v in rng.uniform(0.5, 3.0, c)]
        cls, kw = "WeightsUniformInferableQuantizer", dict(num_bits=5, min_range=lo, max_range=hi, per_channel=True, channel_axis=axis)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        q = getattr(Q, cls)(**kw)
    want = _oracle(cls, kw, x_np, dt)
    native.set_tuning("rowsteps", 1)
    try:
        y = q(x)
        assert "rowsteps_kernel" in native.last_launch(), native.last_launch()
        got = y.float().cpu().numpy()
        assert y.dtype == TORCH[dt] and bits_equal(got, want), first_mismatch(got, want, x_np)
        native.set_tuning("rowsteps", 0)
        y0 = q(x)
        assert "rowsteps_kernel" not in native.last_launch()
        assert torch.equal(y0, y)
    finally:
        native.set_tuning("rowsteps", 2)


def test_rows_of_four_steps_or_ragged_rows_keep_their_kernels():
    q = Q.WeightsSymmetricInferableQuantizer(8, [1.0] * 64, True, 0)
    native.set_tuning("rowsteps", 1)
    try:
        for cols, name in ((4096, "rows_kernel"), (1000, "gather_kernel"), (1024, "gather_kernel"), (2048, "rowsteps_kernel")):
            q(torch.randn(64, cols, device="cuda"))
            assert native.last_launch().startswith(name), (cols, native.last_launch())
    finally:
        native.set_tuning("rowsteps", 2)
    q(torch.randn(64, 2048, device="cuda"))
    assert native.last_launch().startswith("rows_kernel")            # the default: 32 four-step blocks are no full round
    q4 = Q.WeightsSymmetricInferableQuantizer(8, [1.0] * 4096, True, 0)
    q4(torch.randn(4096, 4096, device="cuda").bfloat16())            # 8192 steps = 2048 blocks = one round of 8 per CU:
    assert native.last_launch().startswith("shortrows_kernel"), native.last_launch()   # 16-bit one-round launches (round 6)
    native.set_tuning("shortrows", 0)
    try:
        q4(torch.randn(4096, 4096, device="cuda").bfloat16())
        assert native.last_launch().startswith("rowsteps_kernel"), native.last_launch()   # ... rowsteps_kernel's window without it
    finally:
        native.set_tuning("shortrows", 1)
    q4(torch.randn(4096, 4096, device="cuda"))
    assert native.last_launch().startswith("rows_kernel")
