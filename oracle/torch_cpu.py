"""Torch-CPU leg of the oracle (TEST INFRASTRUCTURE; see mctq_oracle.py's header).

Runs, on CPU tensors, the same third-party ATen operators the reference calls
(torch.fake_quantize_per_{tensor,channel}_affine; the op chain of quantizer_utils.py:121-139),
with parameters derived by the numpy restatement in mctq_oracle.py.  It is the CPU baseline
bench.py reports ("port": the reference's own Python shim cannot travel to the GPU machine)
and an on-box cross-check of the numpy oracle.
"""
from __future__ import annotations

import numpy as np
import torch

from . import mctq_oracle as O


def prepare(cls_name: str, kwargs: dict):
    """Return a callable f(x_cpu_tensor) -> y for a reference quantizer class name + kwargs."""
    kw = dict(kwargs)
    nb = kw["num_bits"]
    if cls_name in ("WeightsSymmetricInferableQuantizer", "WeightsPOTInferableQuantizer",
                    "WeightsUniformInferableQuantizer"):
        if cls_name == "WeightsUniformInferableQuantizer":
            s, z, qmin, qmax, _, _ = O.weights_uniform_params(nb, kw["min_range"], kw["max_range"])
        else:
            s, z, qmin, qmax = O.weights_symmetric_params(nb, kw["threshold"])
        st, zt = torch.from_numpy(s), torch.from_numpy(z)
        if kw["per_channel"]:
            axis = kw["channel_axis"]
            return lambda x: torch.fake_quantize_per_channel_affine(x, st, zt, axis, qmin, qmax)
        return lambda x: torch.fake_quantize_per_tensor_affine(x, st, zt, qmin, qmax)
    if cls_name in ("ActivationSymmetricInferableQuantizer", "ActivationPOTInferableQuantizer"):
        s, z, qmin, qmax = O.activation_symmetric_params(nb, kw["threshold"], kw["signed"])
        return lambda x: torch.fake_quantize_per_tensor_affine(x, s, z, qmin, qmax)
    if cls_name == "ActivationUniformInferableQuantizer":
        s, z, qmin, qmax, _, _ = O.activation_uniform_params(nb, kw["min_range"], kw["max_range"])
        return lambda x: torch.fake_quantize_per_tensor_affine(x, s, z, qmin, qmax)
    if "LUT" in cls_name or "Lut" in cls_name:
        signed = kw.get("signed", True)
        B = kw.get("lut_values_bitwidth", O.LUT_VALUES_BITWIDTH)
        eps = kw.get("eps", O.EPS)
        lut = torch.tensor(kw["lut_values"], dtype=torch.float32)
        mult = 2 ** (B - int(signed))
        cmin, cmax = (-2 ** (B - 1), 2 ** (B - 1) - 1) if signed else (0, 2 ** B - 1)
        if cls_name == "ActivationLutPOTInferableQuantizer":
            thr = float(kw["threshold"][0])
        else:
            thr = torch.from_numpy(np.asarray(kw["threshold"]).astype(np.float32))

        def run(x):
            t = thr
            if isinstance(t, torch.Tensor) and kw.get("per_channel"):
                shape = [1] * x.dim()
                shape[kw["channel_axis"]] = -1
                t = t.reshape(shape)
            v = torch.clip((x / (t + eps)) * mult, min=cmin, max=cmax).unsqueeze(-1)
            idx = torch.argmin(torch.abs(v - lut.reshape([1] * (v.dim() - 1) + [-1])), dim=-1)
            return (lut[idx] / mult) * t
        return run
    raise KeyError(cls_name)
