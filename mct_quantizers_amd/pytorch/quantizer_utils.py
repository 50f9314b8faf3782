"""Host-side helpers of the quantizers (constructor time only).

Same public names and semantics as mct_quantizers/pytorch/quantizer_utils.py:23-92;
``lut_quantizer`` keeps the reference's signature but runs the fused gfx950 kernel on GPU tensors.
"""
from typing import Tuple

import numpy as np
import torch

from mct_quantizers_amd.hip import ops
from mct_quantizers_amd.logger import Logger


def get_working_device() -> torch.device:
    """'cuda' (the HIP device on PyTorch-ROCm) when a GPU is visible, else 'cpu'."""
    return torch.device('cuda' if torch.cuda.is_available() else 'cpu')


def to_torch_tensor(tensor):
    """numpy array / float / int / (nested) list -> tensor on the working device; arrays become float32."""
    dev = get_working_device()
    if isinstance(tensor, torch.Tensor):
        return tensor.to(dev)
    if isinstance(tensor, list):
        return [to_torch_tensor(t) for t in tensor]
    if isinstance(tensor, tuple):
        return (to_torch_tensor(t) for t in tensor)
    if isinstance(tensor, np.ndarray):
        return torch.from_numpy(tensor.astype(np.float32)).to(dev)
    if isinstance(tensor, float):
        return torch.Tensor([tensor]).to(dev)
    if isinstance(tensor, int):
        return torch.Tensor([tensor]).int().to(dev)
    raise Exception(f'Conversion of type {type(tensor)} to {type(torch.Tensor)} is not supported')


def fix_range_to_include_zero(range_min: torch.Tensor, range_max: torch.Tensor,
                              n_bits: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """Move (min, max) so that 0.0 lies on the n_bits quantization grid (float32 tensor math).

    Ranges that straddle zero are shifted by a sub-step amount, all-positive ranges get min=0 and
    all-negative ranges get max=0 (reference pytorch/quantizer_utils.py:73-92).
    """
    straddles = ((range_min <= 0) & (range_max >= 0)).float()
    above = (range_min > 0).float()
    below = (range_max < 0).float()

    step = (range_max - range_min) / (2 ** n_bits - 1)
    shifted_min = step * torch.round(range_min / step)
    shifted_max = range_max - range_min + shifted_min

    new_min = shifted_min * straddles + below * range_min
    new_max = shifted_max * straddles + above * range_max

    span = range_max - range_min
    if not torch.all(torch.isclose((new_min - range_min) / span, torch.tensor(0., device=span.device), atol=1e-6)):
        Logger.warning(f"Adjusting (min_range, max_range) from ({range_min},{range_max}) to ({new_min},{new_max})")
    return new_min, new_max


def lut_quantizer(tensor_data: torch.Tensor, lut_values: torch.Tensor, signed: bool, threshold,
                  lut_values_bitwidth: int, eps: float, per_channel: bool = None, channel_axis: int = None,
                  input_rank: int = None) -> torch.Tensor:
    """Codebook quantization: scale by the threshold into the integer range, snap to the nearest
    codebook entry (first minimum in list order), scale back.

    ``threshold`` is a float32 tensor (one entry, or one per channel) or a Python float.
    """
    mult, cmin, cmax = lut_domain(lut_values_bitwidth, signed)
    if per_channel:
        rank = tensor_data.dim() if hasattr(tensor_data, "dim") else input_rank
        axis = channel_axis % rank if rank else channel_axis      # the reference's reshape accepts negative axes
        return ops.lut_per_channel(tensor_data, lut_values, threshold.reshape(-1), float(eps), axis,
                                   mult, cmin, cmax)
    if isinstance(threshold, torch.Tensor):
        thr = np.float32(threshold.reshape(-1)[0].item())
        thr_div = float(thr + np.float32(eps))          # float32 add, as tensor + python scalar
    else:
        thr = np.float32(threshold)
        thr_div = float(np.float32(float(threshold) + eps))   # double add, then float32
        if getattr(tensor_data, "dtype", None) is torch.float64:
            return ops.lut_per_tensor(tensor_data, lut_values, thr_div, float(thr), mult, cmin, cmax, None, 0,
                                      float(threshold) + eps)
    return ops.lut_per_tensor(tensor_data, lut_values, thr_div, float(thr), mult, cmin, cmax)


def lut_domain(lut_values_bitwidth: int, signed: bool):
    """(multiplier 2^(B-signed), clip_min, clip_max) of int_quantization_with_threshold (:142-170)."""
    mult = float(2 ** (lut_values_bitwidth - int(signed)))
    if signed:
        return mult, float(-2 ** (lut_values_bitwidth - 1)), float(2 ** (lut_values_bitwidth - 1) - 1)
    return mult, 0.0, float(2 ** lut_values_bitwidth - 1)
