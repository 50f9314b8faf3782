"""ONNX-export branch of the quantizers: ``quantizer.enable_custom_impl()`` + ``torch.onnx.export``.

While an export traces the model (``self._use_custom_impl and torch.jit.is_tracing()``), every quantizer
call goes through a ``torch.autograd.Function`` whose ``symbolic`` emits one ``mct_quantizers::<Name>``
node (same domain, op names, attributes and constant inputs as the reference, so the exported file loads
with the reference's onnxruntime ops) and whose ``forward`` computes the reference's export-time
arithmetic -- clip, true division, round, scale back: NOT the fake-quant contract of the normal path.
On a GPU tensor that arithmetic is the gfx950 kernel behind ``mctq_grid_per_*_f32`` (include/mctq_hip.h).

Mirrors, relative to /root/reference/mct_quantizers/pytorch/quantizers/:
  base_quantizer_autograd_function.py:22-59, weights_inferable_quantizers/base_weight_quantizer_autograd_function.py,
  activation_inferable_quantizers/base_activation_quantizer_autograd_function.py
  weights_inferable_quantizers/weights_symmetric_inferable_quantizer.py:32-70,159-215   (WeightsSymmetricF)
  weights_inferable_quantizers/weights_pot_inferable_quantizer.py:99-156                (WeightsPOTF)
  weights_inferable_quantizers/weights_uniform_inferable_quantizer.py:34-78,174-235     (WeightsUniformF)
  weights_inferable_quantizers/weights_lut_symmetric_inferable_quantizer.py:131-205     (WeightsLUTSymmetricF)
  weights_inferable_quantizers/weights_lut_pot_inferable_quantizer.py:106-183           (WeightsLUTPOTF)
  activation_inferable_quantizers/activation_symmetric_inferable_quantizer.py:29-54,120-166   (ActivationSymF)
  activation_inferable_quantizers/activation_pot_inferable_quantizer.py:76-124                (ActivationPOTF)
  activation_inferable_quantizers/activation_uniform_inferable_quantizer.py:32-65,131-177     (ActivationUniformF)
The numpy twins used by the onnxruntime custom ops are CPU export tooling and stay out of scope.
"""
from typing import Any, Dict

import numpy as np
import torch

from mct_quantizers_amd.common.constants import MCTQ_VERSION, ONNX_CUSTOM_OP_DOMAIN, REFERENCE_API_VERSION
from mct_quantizers_amd.hip import ops
from mct_quantizers_amd.pytorch.quantizer_utils import fix_range_to_include_zero, lut_quantizer, to_torch_tensor


def is_export_tracing(quantizer) -> bool:
    """The reference's switch into this branch."""
    return quantizer._use_custom_impl and torch.jit.is_tracing()


# ------------------------------------------------------------------------------------------
# export-time arithmetic (parameters on the host, float32 / double exactly as the reference's CPU run)
# ------------------------------------------------------------------------------------------

def _host_f32(values) -> torch.Tensor:
    return torch.from_numpy(np.asarray(values, dtype=np.float64).astype(np.float32).reshape(-1))


def quantize_sym_weights_torch(input_tensor: torch.Tensor, num_bits: int, threshold, per_channel: bool,
                               channel_axis: int) -> torch.Tensor:
    """Symmetric weights on the export grid: lo = -thr, hi = thr - step, step = thr / 2^(num_bits-1) (float32)."""
    thr = _host_f32(threshold)
    step = thr / (2 ** (num_bits - 1))
    lo, hi = -thr, thr - step
    if per_channel:
        return ops.grid_per_channel(input_tensor, lo, hi, step, channel_axis % input_tensor.dim())
    return ops.grid_per_tensor(input_tensor, lo.item(), hi.item(), step.item())


def quantize_uniform_weights_torch(input_tensor: torch.Tensor, num_bits: int, min_range, max_range,
                                   per_channel: bool, channel_axis: int = None) -> torch.Tensor:
    """Uniform weights on the export grid; note: rint(c / step) * step, the grid is NOT shifted by the minimum."""
    lo, hi = fix_range_to_include_zero(_host_f32(min_range), _host_f32(max_range), num_bits)
    step = (hi - lo) / (2 ** num_bits - 1)
    if per_channel:
        return ops.grid_per_channel(input_tensor, lo, hi, step, channel_axis % input_tensor.dim())
    return ops.grid_per_tensor(input_tensor, lo.item(), hi.item(), step.item())


def quantize_sym_activations_torch(input_tensor: torch.Tensor, threshold: float, signed: bool,
                                   num_bits: int) -> torch.Tensor:
    """Symmetric activations: parameters in double, narrowed to float32 where they meet the tensor."""
    threshold = float(threshold)
    if signed:
        step = threshold / (2 ** (num_bits - 1))
        lo, hi = -threshold, threshold - step
    else:
        step = threshold / (2 ** num_bits)
        lo, hi = 0.0, threshold - step
    if lo > hi:                 # torch.clip with min > max yields max everywhere
        lo = hi
    return ops.grid_per_tensor(input_tensor, lo, hi, step)


def adjust_range_to_include_zero(range_min, range_max, n_bits: int):
    """numpy (double) twin of fix_range_to_include_zero, plus a final clamp to lo <= 0 <= hi
    (common/quant_utils.py:20-50).  Written with the same 0/1 masks so that signed zeros come out alike."""
    range_min, range_max = np.float64(range_min), np.float64(range_max)
    step = (range_max - range_min) / (2 ** n_bits - 1)
    shifted_lo = step * np.round(range_min / step)
    shifted_hi = range_max - range_min + shifted_lo
    above, below = range_min > 0, range_max < 0
    straddles = np.logical_and(np.logical_not(above), np.logical_not(below))
    lo = shifted_lo * straddles + below * range_min
    hi = shifted_hi * straddles + above * range_max
    return float(np.minimum(lo, 0)), float(np.maximum(hi, 0))


def quantize_uniform_activations_torch(tensor_data: torch.Tensor, range_min: float, range_max: float,
                                       n_bits: int) -> torch.Tensor:
    """Uniform activations: step * rint((c - lo) / step) + lo."""
    lo, hi = adjust_range_to_include_zero(range_min, range_max, n_bits)
    step = (hi - lo) / (2 ** n_bits - 1)
    return ops.grid_per_tensor(tensor_data, lo, hi, step, shifted=True)


# ------------------------------------------------------------------------------------------
# autograd functions: forward = the arithmetic above, symbolic = the ONNX node
# ------------------------------------------------------------------------------------------

class BaseQuantizerAutogradFunction(torch.autograd.Function):
    """forward + symbolic only; there is no backward for inference-time quantizers."""

    @staticmethod
    def forward(ctx, input_tensor, **kwargs):
        raise NotImplementedError

    @staticmethod
    def symbolic(g, input_tensor, **kwargs):
        raise NotImplementedError

    def backward(ctx: Any, *grad_outputs: Any) -> Any:
        raise NotImplementedError()

    @staticmethod
    def _get_metadata_attributes() -> Dict[str, Any]:
        # the op-set version the exported nodes are written against
        return {f"{MCTQ_VERSION}_s": REFERENCE_API_VERSION}


class BaseWeightQuantizerAutogradFunction(BaseQuantizerAutogradFunction):
    @staticmethod
    def is_signed():
        return True


class BaseActivationQuantizerAutogradFunction(BaseQuantizerAutogradFunction):
    pass


def _const_f32(g, values):
    return g.op('Constant', value_t=torch.tensor(values, dtype=torch.float32))


def _weights_threshold_node(g, cls, op_name, input_tensor, num_bits, threshold, per_channel, channel_axis):
    if not per_channel and channel_axis is None:
        channel_axis = 0        # the onnxruntime op needs the attribute to exist
    return g.op(f"{ONNX_CUSTOM_OP_DOMAIN}::{op_name}", input_tensor, _const_f32(g, threshold),
                num_bits_i=num_bits, per_channel_i=int(per_channel), channel_axis_i=channel_axis,
                signed_i=int(cls.is_signed()), **cls._get_metadata_attributes()).setType(input_tensor.type())


class WeightsSymmetricF(BaseWeightQuantizerAutogradFunction):
    @staticmethod
    def forward(ctx, input_tensor, num_bits, threshold, per_channel, channel_axis):
        return quantize_sym_weights_torch(input_tensor, num_bits, threshold, per_channel, channel_axis)

    @staticmethod
    def symbolic(g, input_tensor, num_bits, threshold, per_channel, channel_axis):
        return _weights_threshold_node(g, WeightsSymmetricF, "WeightsSymmetricQuantizer", input_tensor, num_bits,
                                       threshold, per_channel, channel_axis)


class WeightsPOTF(BaseWeightQuantizerAutogradFunction):
    @staticmethod
    def forward(ctx, input_tensor, num_bits, threshold, per_channel, channel_axis):
        return quantize_sym_weights_torch(input_tensor, num_bits, threshold, per_channel, channel_axis)

    @staticmethod
    def symbolic(g, input_tensor, num_bits, threshold, per_channel, channel_axis):
        return _weights_threshold_node(g, WeightsPOTF, "WeightsPOTQuantizer", input_tensor, num_bits, threshold,
                                       per_channel, channel_axis)


class WeightsUniformF(BaseWeightQuantizerAutogradFunction):
    @staticmethod
    def forward(ctx, input_tensor, num_bits, min_range, max_range, per_channel, channel_axis):
        return quantize_uniform_weights_torch(input_tensor, num_bits, min_range, max_range, per_channel, channel_axis)

    @staticmethod
    def symbolic(g, input_tensor, num_bits, min_range, max_range, per_channel, channel_axis):
        if not per_channel and channel_axis is None:
            channel_axis = 0
        return g.op(f"{ONNX_CUSTOM_OP_DOMAIN}::WeightsUniformQuantizer", input_tensor,
                    _const_f32(g, min_range), _const_f32(g, max_range),
                    num_bits_i=num_bits, per_channel_i=int(per_channel), channel_axis_i=channel_axis,
                    signed_i=WeightsUniformF.is_signed(),
                    **WeightsUniformF._get_metadata_attributes()).setType(input_tensor.type())


def _lut_forward(input_tensor, lut_values, threshold, lut_values_bitwidth, eps, per_channel, channel_axis, input_rank):
    # the symbolic needs numpy arrays, the arithmetic needs tensors next to the input
    return lut_quantizer(input_tensor, lut_values=to_torch_tensor(lut_values).to(input_tensor.device), signed=True,
                         threshold=to_torch_tensor(threshold).to(input_tensor.device),
                         lut_values_bitwidth=lut_values_bitwidth, eps=eps, per_channel=per_channel,
                         channel_axis=channel_axis, input_rank=input_rank)


def _lut_node(g, cls, op_name, input_tensor, num_bits, lut_values, threshold, lut_values_bitwidth, eps, per_channel,
              channel_axis, input_rank):
    if not per_channel:
        if channel_axis is None:
            channel_axis = 0
        if input_rank is None:
            input_rank = 4
    return g.op(f"{ONNX_CUSTOM_OP_DOMAIN}::{op_name}", input_tensor, _const_f32(g, lut_values),
                _const_f32(g, threshold),
                num_bits_i=num_bits, per_channel_i=int(per_channel), channel_axis_i=channel_axis,
                input_rank_i=input_rank, lut_values_bitwidth_i=lut_values_bitwidth, eps_f=eps,
                signed_i=int(cls.is_signed()), **cls._get_metadata_attributes()).setType(input_tensor.type())


class WeightsLUTSymmetricF(BaseWeightQuantizerAutogradFunction):
    @staticmethod
    def forward(ctx, input_tensor, num_bits, lut_values, threshold, lut_values_bitwidth, eps, per_channel,
                channel_axis, input_rank):
        return _lut_forward(input_tensor, lut_values, threshold, lut_values_bitwidth, eps, per_channel, channel_axis,
                            input_rank)

    @staticmethod
    def symbolic(g, input_tensor, num_bits, lut_values, threshold, lut_values_bitwidth, eps, per_channel,
                 channel_axis, input_rank):
        return _lut_node(g, WeightsLUTSymmetricF, "WeightsLUTSymmetricQuantizer", input_tensor, num_bits, lut_values,
                         threshold, lut_values_bitwidth, eps, per_channel, channel_axis, input_rank)


class WeightsLUTPOTF(BaseWeightQuantizerAutogradFunction):
    @staticmethod
    def forward(ctx, input_tensor, num_bits, lut_values, threshold, lut_values_bitwidth, eps, per_channel,
                channel_axis, input_rank):
        return _lut_forward(input_tensor, lut_values, threshold, lut_values_bitwidth, eps, per_channel, channel_axis,
                            input_rank)

    @staticmethod
    def symbolic(g, input_tensor, num_bits, lut_values, threshold, lut_values_bitwidth, eps, per_channel,
                 channel_axis, input_rank):
        return _lut_node(g, WeightsLUTPOTF, "WeightsLUTPOTQuantizer", input_tensor, num_bits, lut_values, threshold,
                         lut_values_bitwidth, eps, per_channel, channel_axis, input_rank)


def _activation_threshold_node(g, cls, op_name, input_tensor, threshold, signed, num_bits):
    return g.op(f"{ONNX_CUSTOM_OP_DOMAIN}::{op_name}", input_tensor, threshold_f=threshold, signed_i=int(signed),
                num_bits_i=num_bits, **cls._get_metadata_attributes()).setType(input_tensor.type())


class ActivationSymF(BaseActivationQuantizerAutogradFunction):
    @staticmethod
    def forward(ctx, input_tensor, threshold, signed, num_bits):
        return quantize_sym_activations_torch(input_tensor, threshold, signed, num_bits)

    @staticmethod
    def symbolic(g, input_tensor, threshold, signed, num_bits):
        return _activation_threshold_node(g, ActivationSymF, "ActivationSymmetricQuantizer", input_tensor, threshold,
                                          signed, num_bits)


class ActivationPOTF(BaseActivationQuantizerAutogradFunction):
    @staticmethod
    def forward(ctx, input_tensor, threshold, signed, num_bits):
        return quantize_sym_activations_torch(input_tensor, threshold, signed, num_bits)

    @staticmethod
    def symbolic(g, input_tensor, threshold, signed, num_bits):
        return _activation_threshold_node(g, ActivationPOTF, "ActivationPOTQuantizer", input_tensor, threshold,
                                          signed, num_bits)


class ActivationUniformF(BaseActivationQuantizerAutogradFunction):
    @staticmethod
    def forward(ctx, input_tensor, min_range, max_range, num_bits):
        return quantize_uniform_activations_torch(input_tensor, min_range, max_range, num_bits)

    @staticmethod
    def symbolic(g, input_tensor, min_range, max_range, num_bits):
        return g.op(f"{ONNX_CUSTOM_OP_DOMAIN}::ActivationUniformQuantizer", input_tensor, min_range_f=min_range,
                    max_range_f=max_range, num_bits_i=num_bits,
                    **ActivationUniformF._get_metadata_attributes()).setType(input_tensor.type())
