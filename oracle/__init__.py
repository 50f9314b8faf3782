"""CPU oracle package (test infrastructure only; see mctq_oracle.py's header).

``oracle_call`` maps a reference quantizer class name + constructor kwargs + input
to the oracle's output, the way the reference's ``Cls(**kwargs)(x)`` would.
"""
from __future__ import annotations

import numpy as np

from . import mctq_oracle as O


def oracle_call(cls_name: str, kwargs: dict, x: np.ndarray, return_index: bool = False, in_dtype: str = "float32"):
    """``x`` holds the tensor's values widened to float32; ``in_dtype`` is the tensor's storage type.
    Affine quantizers return values of that type (widened), LUT quantizers return float32."""
    if in_dtype == "float64":
        assert not return_index
        kw = dict(kwargs)
        nb = kw["num_bits"]
        axis = kw.get("channel_axis") if kw.get("per_channel") else None
        if cls_name in ("WeightsSymmetricInferableQuantizer", "WeightsPOTInferableQuantizer"):
            s, z, qmin, qmax = O.weights_symmetric_params(nb, kw["threshold"])
        elif cls_name == "WeightsUniformInferableQuantizer":
            s, z, qmin, qmax, _, _ = O.weights_uniform_params(nb, kw["min_range"], kw["max_range"])
        elif cls_name in ("ActivationSymmetricInferableQuantizer", "ActivationPOTInferableQuantizer"):
            s, z, qmin, qmax = O.activation_symmetric_params(nb, kw["threshold"], kw["signed"])
        elif cls_name == "ActivationUniformInferableQuantizer":
            s, z, qmin, qmax, _, _ = O.activation_uniform_params(nb, kw["min_range"], kw["max_range"])
        elif cls_name in ("WeightsLUTSymmetricInferableQuantizer", "WeightsLUTPOTInferableQuantizer"):
            return O.lut_quantize_f64(x, kw["lut_values"], np.asarray(kw["threshold"], dtype=np.float64).astype(np.float32),
                                      True, kw.get("lut_values_bitwidth", O.LUT_VALUES_BITWIDTH), kw.get("eps", O.EPS),
                                      per_channel=kw["per_channel"], channel_axis=kw.get("channel_axis"))
        elif cls_name == "ActivationLutPOTInferableQuantizer":
            return O.lut_quantize_f64(x, kw["lut_values"], float(kw["threshold"][0]), kw["signed"],
                                      kw.get("lut_values_bitwidth", O.LUT_VALUES_BITWIDTH), kw.get("eps", O.EPS))
        else:
            raise KeyError(cls_name)
        return O.fake_quant_affine_f64(x, s, z, qmin, qmax, axis=axis)
    if in_dtype != "float32":
        assert not return_index
        if "LUT" in cls_name or "Lut" in cls_name:
            if cls_name == "ActivationLutPOTInferableQuantizer":
                kw = dict(kwargs)
                return O.lut_quantize(x, kw["lut_values"], float(kw["threshold"][0]), kw["signed"],
                                      kw.get("lut_values_bitwidth", O.LUT_VALUES_BITWIDTH), kw.get("eps", O.EPS),
                                      step_dtype=in_dtype)
            return oracle_call(cls_name, kwargs, x)
        return O.narrow(oracle_call(cls_name, kwargs, x), in_dtype)
    kw = dict(kwargs)
    nb = kw["num_bits"]
    if cls_name in ("WeightsSymmetricInferableQuantizer", "WeightsPOTInferableQuantizer"):
        s, z, qmin, qmax = O.weights_symmetric_params(nb, kw["threshold"])
        axis = kw.get("channel_axis") if kw["per_channel"] else None
        return O.fake_quant_affine(x, s, z, qmin, qmax, axis=axis, return_index=return_index)
    if cls_name == "WeightsUniformInferableQuantizer":
        s, z, qmin, qmax, _, _ = O.weights_uniform_params(nb, kw["min_range"], kw["max_range"])
        axis = kw.get("channel_axis") if kw["per_channel"] else None
        return O.fake_quant_affine(x, s, z, qmin, qmax, axis=axis, return_index=return_index)
    if cls_name in ("ActivationSymmetricInferableQuantizer", "ActivationPOTInferableQuantizer"):
        s, z, qmin, qmax = O.activation_symmetric_params(nb, kw["threshold"], kw["signed"])
        return O.fake_quant_affine(x, s, z, qmin, qmax, return_index=return_index)
    if cls_name == "ActivationUniformInferableQuantizer":
        s, z, qmin, qmax, _, _ = O.activation_uniform_params(nb, kw["min_range"], kw["max_range"])
        return O.fake_quant_affine(x, s, z, qmin, qmax, return_index=return_index)
    if cls_name in ("WeightsLUTSymmetricInferableQuantizer", "WeightsLUTPOTInferableQuantizer"):
        return O.lut_quantize(x, kw["lut_values"], np.asarray(kw["threshold"], dtype=np.float64).astype(np.float32),
                              True, kw.get("lut_values_bitwidth", O.LUT_VALUES_BITWIDTH), kw.get("eps", O.EPS),
                              per_channel=kw["per_channel"], channel_axis=kw.get("channel_axis"),
                              return_index=return_index)
    if cls_name == "ActivationLutPOTInferableQuantizer":
        return O.lut_quantize(x, kw["lut_values"], float(kw["threshold"][0]), kw["signed"],
                              kw.get("lut_values_bitwidth", O.LUT_VALUES_BITWIDTH), kw.get("eps", O.EPS),
                              return_index=return_index)
    raise KeyError(cls_name)


def oracle_export_call(cls_name: str, kwargs: dict, x: np.ndarray):
    """What the reference's ``q.enable_custom_impl(); q(x)`` returns while ``torch.jit`` is tracing."""
    kw = dict(kwargs)
    nb = kw["num_bits"]
    axis = kw.get("channel_axis") if kw.get("per_channel") else None
    if axis is not None:
        axis %= x.ndim
    if cls_name in ("WeightsSymmetricInferableQuantizer", "WeightsPOTInferableQuantizer"):
        return O.export_weights_symmetric(x, nb, kw["threshold"], axis)
    if cls_name == "WeightsUniformInferableQuantizer":
        _, _, _, _, a, b = O.weights_uniform_params(nb, kw["min_range"], kw["max_range"])
        return O.export_weights_uniform(x, nb, a, b, axis)
    if cls_name in ("ActivationSymmetricInferableQuantizer", "ActivationPOTInferableQuantizer"):
        return O.export_activation_symmetric(x, nb, float(np.asarray(kw["threshold"])[0]), kw["signed"])
    if cls_name == "ActivationUniformInferableQuantizer":
        _, _, _, _, a, b = O.activation_uniform_params(nb, kw["min_range"], kw["max_range"])
        return O.export_activation_uniform(x, nb, a, b)
    if cls_name in ("WeightsLUTSymmetricInferableQuantizer", "WeightsLUTPOTInferableQuantizer"):
        return oracle_call(cls_name, kwargs, x)        # the LUT export branch runs the same chain
    raise KeyError(cls_name)
