"""LUT (codebook) inferable quantizers: weights LUT-symmetric / LUT-POT and activation LUT-POT.

Public contract mirrored from, relative to /root/reference/mct_quantizers/pytorch/quantizers/:
  base_lut_symmetric_inferable_quantizer.py:32-94               validation + attributes
  weights_inferable_quantizers/weights_lut_symmetric_inferable_quantizer.py:42-128
  weights_inferable_quantizers/weights_lut_pot_inferable_quantizer.py:40-104
  activation_inferable_quantizers/activation_lut_pot_inferable_quantizer.py:38-91

``__call__`` replaces the reference's ~10-kernel op chain (quantizer_utils.py:95-139, two N x L
temporaries) by one fused gfx950 kernel at 8 algorithmic bytes per element.
"""
import warnings
from typing import List

import numpy as np
import torch

from mct_quantizers_amd.common.constants import EPS, LUT_VALUES_BITWIDTH
from mct_quantizers_amd.common.registry import QuantizationMethod, QuantizationTarget, QuantizerID, mark_quantizer
from mct_quantizers_amd.hip import native, ops
from mct_quantizers_amd.pytorch.quantizer_utils import get_working_device, lut_domain, to_torch_tensor
from mct_quantizers_amd.pytorch.quantizers.affine import BasePyTorchInferableQuantizer, _is_compiling, _is_pot


@mark_quantizer(quantization_target=None,
                quantization_method=[QuantizationMethod.LUT_SYM_QUANTIZER],
                identifier=QuantizerID.INFERABLE)
class BaseLUTSymmetricInferableQuantizer(BasePyTorchInferableQuantizer):

    # The reference reads these attributes on every call (weights_lut_symmetric_inferable_quantizer.py:112-121,
    # activation_lut_pot_inferable_quantizer.py:86-91); here the launch state (divisors, decision table / threshold list,
    # pre-packed launch) is derived from them once.  Assigning to one of them after construction marks that state stale
    # and the next call re-derives it, so `q.threshold = 8.0` or `q.eps = 0.0` take effect exactly as in the reference.
    # (In-place edits of a codebook TENSOR are not seen: replace the attribute instead.)
    _launch_inputs = frozenset(("threshold", "lut_values", "signed", "lut_values_bitwidth", "eps",
                                "_threshold_torch", "_lut_values_torch"))

    def __setattr__(self, name, value):
        object.__setattr__(self, name, value)
        if name in self._launch_inputs and self.__dict__.get("_launch_ready"):
            self.__dict__["_stale"] = True

    def _resync(self):
        """Re-derive the launch state from the attributes the reference reads at call time."""
        d = self.__dict__
        d["_launch_ready"] = False

        def host(v):
            if isinstance(v, torch.Tensor):
                return v.detach().cpu().numpy()
            return np.asarray(v if isinstance(v, (list, tuple, np.ndarray)) else [v])
        if "_threshold_torch" in d:                      # weights classes: the private tensors are what is read
            d["_threshold_np"], d["_lut_values_np"] = host(d["_threshold_torch"]), host(d["_lut_values_torch"])
        else:                                            # activation class: public threshold (float) and lut_values (tensor)
            d["_threshold_np"], d["_lut_values_np"] = host(self.threshold), host(self.lut_values)
        self._rebuild_launch_state()

    def __init__(self, num_bits: int, lut_values: List[float], threshold: List[float], signed: bool,
                 lut_values_bitwidth: int, eps: float):
        super().__init__()
        assert isinstance(threshold, list), f'Threshold is expected to be a list, but is of type {type(threshold)}'
        assert isinstance(lut_values, list), f'lut_values is expected to be a list, but is of type {type(lut_values)}'

        self._threshold_np = np.asarray(threshold)
        self._lut_values_np = np.asarray(lut_values)
        lut = self._lut_values_np

        assert len(np.unique(lut)) <= 2 ** num_bits, \
            f'Expected num of lut values to be less or equal than {2 ** num_bits} but got {len(lut)}'
        assert not np.any(lut - lut.astype(int)), 'Expected lut values to be integers'
        if signed:
            bound = 2 ** (lut_values_bitwidth - 1)
            assert np.all((-bound <= lut) & (lut <= bound - 1)), 'Expected lut values in the quantization range'
        else:
            assert np.all(lut <= 2 ** lut_values_bitwidth), 'Expected lut values in the quantization range'
            assert np.all(lut >= 0), 'Expected unsigned lut values in unsigned activation quantization'
        assert num_bits <= lut_values_bitwidth, \
            f'Look-Up-Table bit configuration has {num_bits} bits. It must be less then {lut_values_bitwidth}'
        if num_bits == lut_values_bitwidth:
            warnings.warn("Num of bits equal to multiplier n bits, Please be aware LUT quantizier may be "
                          "inefficient in that case, consider using SymmetricInferableQuantizer instead")

        self.threshold = threshold
        self.lut_values = lut_values
        self.signed = signed
        self.num_bits = num_bits
        self.lut_values_bitwidth = lut_values_bitwidth
        self.eps = eps


@mark_quantizer(quantization_target=QuantizationTarget.Weights,
                quantization_method=[QuantizationMethod.LUT_SYM_QUANTIZER],
                identifier=QuantizerID.INFERABLE)
class WeightsLUTSymmetricInferableQuantizer(BaseLUTSymmetricInferableQuantizer):
    """Signed codebook quantizer for weights, one threshold per tensor or per channel."""

    def __init__(self, num_bits: int, lut_values: List[float], threshold: List[float], per_channel: bool,
                 channel_axis: int = None, input_rank: int = None, lut_values_bitwidth: int = LUT_VALUES_BITWIDTH,
                 eps: float = EPS):
        super().__init__(threshold=threshold, num_bits=num_bits, lut_values=lut_values, signed=True,
                         lut_values_bitwidth=lut_values_bitwidth, eps=eps)
        self.per_channel = per_channel
        self.channel_axis = channel_axis
        self.input_rank = input_rank
        if per_channel:
            assert channel_axis is not None, 'Channel axis is missing in per channel quantization'
            assert input_rank is not None, 'input_rank is missing in per channel quantization'
            assert len(threshold) >= 1, \
                f'In per-channel quantization threshold should be of length >= 1 but is {len(threshold)}'
        else:
            assert len(threshold) == 1, \
                f'In per-tensor quantization threshold should be of length 1 but is {len(threshold)}'

        self._rebuild_launch_state()

    def _rebuild_launch_state(self):
        self.__dict__["_launch_ready"] = False           # (our own assignments below are not "changes")
        dev = get_working_device()
        self._threshold_torch = to_torch_tensor(self._threshold_np).to(dev)
        self._lut_values_torch = to_torch_tensor(self._lut_values_np).to(dev)
        # host copies for the per-tensor launch: tensor + python-scalar is a float32 add
        thr0 = np.float32(self._threshold_np.reshape(-1)[0])
        self._thr_mul0 = float(thr0)
        self._thr_div0 = float(thr0 + np.float32(self.eps))
        # codebook compiled once into an LDS decision table (None -> literal scan kernels)
        self._lut_table_torch = ops.make_lut_table(self._lut_values_np, *lut_domain(self.lut_values_bitwidth, True),
                                                   dev)
        # ... or, for clip ranges too wide for the table, into a sorted threshold list (None -> literal scan)
        self._lut_steps_torch = None if self._lut_table_torch is not None else \
            ops.make_lut_steps(self._lut_values_np, *lut_domain(self.lut_values_bitwidth, True), dev)
        self.__dict__["_stale"] = False
        self.__dict__["_launch_ready"] = True

    _export_function = "WeightsLUTSymmetricF"

    def batch_item_lut(self, inputs: torch.Tensor):
        """This quantizer's call on ``inputs`` as one entry of a batched LUT launch (``BatchPlan`` item
        ("lut", x, y, thresholds | None, table, axis | None, eps, thr_div, thr_mul, mult, clip_min, clip_max, step_round) with
        y left as None for the caller to fill in), or None when the call cannot ride a batched launch (no decision
        table for this codebook, float64 or non-contiguous tensor, rank mismatch)."""
        if self.__dict__.get("_stale"):
            self._resync()
        table = self._lut_table_torch
        if (table is None or type(inputs) not in (torch.Tensor, torch.nn.Parameter) or not inputs.is_cuda
                or inputs.dtype not in (torch.float32, torch.float16, torch.bfloat16) or not inputs.is_contiguous()):
            return None
        mult, cmin, cmax = lut_domain(self.lut_values_bitwidth, True)
        if self.per_channel:
            if self.input_rank != inputs.dim():
                return None
            thr = self._threshold_torch
            if thr.dim() != 1 or not thr.is_contiguous():
                return None
            return ("lut", inputs, None, thr, table, self.channel_axis % inputs.dim(), float(self.eps), 0.0, 0.0,
                    mult, cmin, cmax, 0)
        return ("lut", inputs, None, None, table, None, 0.0, self._thr_div0, self._thr_mul0, mult, cmin, cmax, 0)

    def __call__(self, inputs: torch.Tensor) -> torch.Tensor:
        if self._cached(inputs):
            return self.resue_outputs
        if self.__dict__.get("_stale"):
            self._resync()
        if self._use_custom_impl and torch.jit.is_tracing():
            from mct_quantizers_amd.pytorch.quantizers import onnx_export
            return self._remember(getattr(onnx_export, self._export_function).apply(
                inputs, self.num_bits, self._lut_values_np, self._threshold_np, self.lut_values_bitwidth, self.eps,
                self.per_channel, self.channel_axis, self.input_rank))
        if inputs.requires_grad:                # (read first: a tensor that already has it off stays traceable by dynamo)
            inputs.requires_grad = False
        mult, cmin, cmax = lut_domain(self.lut_values_bitwidth, True)
        if self.per_channel:
            if self.input_rank != inputs.dim():
                raise RuntimeError(f'input_rank={self.input_rank} does not match a tensor of rank {inputs.dim()}')
            axis = self.channel_axis % inputs.dim()
            out = ops.lut_per_channel(inputs, self._lut_values_torch, self._threshold_torch, float(self.eps), axis,
                                      mult, cmin, cmax, self._lut_table_torch, self.__dict__.get("_lut_steps_torch"))
        else:
            out = ops.lut_per_tensor(inputs, self._lut_values_torch, self._thr_div0, self._thr_mul0, mult, cmin, cmax,
                                     self._lut_table_torch, steps=self.__dict__.get("_lut_steps_torch"))
        return self._remember(out)


@mark_quantizer(quantization_target=QuantizationTarget.Weights,
                quantization_method=[QuantizationMethod.LUT_POT_QUANTIZER],
                identifier=QuantizerID.INFERABLE)
class WeightsLUTPOTInferableQuantizer(WeightsLUTSymmetricInferableQuantizer):
    """Weights codebook quantizer whose thresholds must be powers of two."""
    _export_function = "WeightsLUTPOTF"

    def __init__(self, num_bits: int, lut_values: List[float], threshold: List[float], per_channel: bool,
                 channel_axis: int = None, input_rank: int = None, lut_values_bitwidth: int = LUT_VALUES_BITWIDTH,
                 eps: float = EPS):
        super().__init__(num_bits=num_bits, threshold=threshold, lut_values=lut_values, per_channel=per_channel,
                         channel_axis=channel_axis, input_rank=input_rank, lut_values_bitwidth=lut_values_bitwidth,
                         eps=eps)
        assert _is_pot(self._threshold_np), f'Expected threshold to be power of 2 but is {threshold}'


def _bounds_in(dt, cmin: float, cmax: float):
    """(clip_min, clip_max) as torch.clip sees them on a tensor of type ``dt``: converted to that type -- or the
    message of the RuntimeError torch raises when a bound does not fit it."""
    if dt is torch.float16 and max(abs(cmin), abs(cmax)) > 65504.0:
        return "value cannot be converted to type c10::Half without overflow"
    lo, hi = (float(torch.tensor(v, dtype=torch.float64).to(dt)) for v in (cmin, cmax))
    return lo, hi


@mark_quantizer(quantization_target=QuantizationTarget.Activation,
                quantization_method=[QuantizationMethod.LUT_POT_QUANTIZER],
                identifier=QuantizerID.INFERABLE)
class ActivationLutPOTInferableQuantizer(BaseLUTSymmetricInferableQuantizer):
    """Activation codebook quantizer (per tensor, power-of-two threshold), signed or unsigned."""

    def __init__(self, num_bits: int, lut_values: List[float], threshold: List[float], signed: bool,
                 lut_values_bitwidth: int = LUT_VALUES_BITWIDTH, eps: float = EPS):
        super().__init__(num_bits=num_bits, lut_values=lut_values, threshold=threshold, signed=signed,
                         lut_values_bitwidth=lut_values_bitwidth, eps=eps)
        assert _is_pot(self._threshold_np), f'Expected threshold to be power of 2 but is {threshold}'
        assert len(self.threshold) == 1, ('For activation, quantization per channel is not supported and threshold '
                                          f'should be of length 1 but is {len(threshold)}')
        self.threshold = self.threshold[0]
        self._rebuild_launch_state()

    def _rebuild_launch_state(self):
        self.__dict__["_launch_ready"] = False           # (our own assignments below are not "changes")
        dev = get_working_device()
        self.lut_values = to_torch_tensor(self._lut_values_np).to(dev)
        # Python-float threshold: threshold + eps is a DOUBLE add; the division then narrows it to the
        # tensor's type (float32, or float16/bfloat16 for half-precision activations).
        self._thr_mul0 = float(np.float32(self.threshold))
        div64 = torch.tensor([float(self.threshold) + self.eps], dtype=torch.float64)
        self._thr_div_by_dtype = {dt: float(div64.to(dt).item())
                                  for dt in (torch.float32, torch.float16, torch.bfloat16)}
        self._lut_table_torch = ops.make_lut_table(self._lut_values_np,
                                                   *lut_domain(self.lut_values_bitwidth, self.signed), dev)
        self._lut_steps_torch = None if self._lut_table_torch is not None else \
            ops.make_lut_steps(self._lut_values_np, *lut_domain(self.lut_values_bitwidth, self.signed), dev)
        # pre-packed launch (compiled binding): activations are launch-bound, see ActivationSymmetric.__call__
        plan = False
        fast = ops._fast_mod() if self._lut_table_torch is not None else None
        mult, cmin, cmax = lut_domain(self.lut_values_bitwidth, self.signed)
        # torch.clip on a half-precision tensor converts the bounds to the tensor's type (quantizer_utils.py:129): bounds
        # that are not exact there (511 -> 512 in bfloat16; 65535 does not fit float16 at all) are another clip range,
        # which the codebook's table / threshold list was not built for
        self._clip_by_dtype = {dt: _bounds_in(dt, cmin, cmax) for dt in (torch.float16, torch.bfloat16)}
        exact = all(b == (cmin, cmax) for b in self._clip_by_dtype.values())
        if fast is not None:
            d = self._thr_div_by_dtype
            plan = fast.LutPlan(self._lut_table_torch, d[torch.float32], d[torch.float16], d[torch.bfloat16],
                                self._thr_mul0, mult, cmin, cmax, 1 if exact else 2)
        self.__dict__["_plan"] = plan
        self.__dict__["_stale"] = False
        self.__dict__["_launch_ready"] = True

    def batch_item_lut(self, inputs: torch.Tensor):
        """As ``WeightsLUTSymmetricInferableQuantizer.batch_item_lut`` for one activation batch (per tensor): the divisor,
        the per-step roundings and the clip range are those of the tensor's own type, as in ``__call__``."""
        if self.__dict__.get("_stale"):
            self._resync()
        if (type(inputs) not in (torch.Tensor, torch.nn.Parameter) or not inputs.is_cuda or not inputs.is_contiguous()
                or inputs.dtype not in (torch.float32, torch.float16, torch.bfloat16)):
            return None
        mult, cmin, cmax = lut_domain(self.lut_values_bitwidth, self.signed)
        dt = inputs.dtype
        table = self._lut_table_torch
        step = {torch.float16: native.DT_F16, torch.bfloat16: native.DT_BF16}.get(dt, 0)
        if step:
            bounds = self.__dict__.get("_clip_by_dtype", {}).get(dt) or _bounds_in(dt, cmin, cmax)
            if isinstance(bounds, str):
                return None                                   # torch.clip raises for this type: let __call__ raise it
            if bounds != (cmin, cmax):
                cmin, cmax = bounds
                table = ops._op_table(self.lut_values, mult, cmin, cmax)[0]
        if table is None:
            return None
        return ("lut", inputs, None, None, table, None, 0.0, self._thr_div_by_dtype[dt], self._thr_mul0, mult, cmin, cmax, step)

    def __call__(self, inputs: torch.Tensor):
        if self.__dict__.get("_stale"):
            self._resync()
        plan = self.__dict__.get("_plan", False)
        if plan is not False and not _is_compiling():
            y = plan(inputs)
            if y is not NotImplemented:
                return y if inputs.is_contiguous() else y.contiguous()      # ops._lut_result
        mult, cmin, cmax = lut_domain(self.lut_values_bitwidth, self.signed)
        dt = getattr(inputs, "dtype", torch.float32)
        step = {torch.float16: native.DT_F16, torch.bfloat16: native.DT_BF16}.get(dt, 0)
        if type(inputs) in (torch.Tensor, torch.nn.Parameter) and not inputs.is_cuda:
            # CPU tensor: hand ATen the same double the reference hands it
            return ops.lut_per_tensor(inputs, self.lut_values, float(self.threshold) + self.eps, self._thr_mul0,
                                      mult, cmin, cmax, None, step or -1)
        thr_div = self._thr_div_by_dtype.get(dt, self._thr_div_by_dtype[torch.float32])
        # a float64 tensor divided by the Python-float threshold + eps: the divisor stays a double
        div64 = float(self.threshold) + self.eps if dt is torch.float64 else None
        if step:
            bounds = self.__dict__.get("_clip_by_dtype", {}).get(dt) or _bounds_in(dt, cmin, cmax)
            if isinstance(bounds, str):
                raise RuntimeError(bounds)                    # what torch.clip raises for this tensor type
            if bounds != (cmin, cmax):                        # the tensor type's own clip range and its own table
                table, steps = ops._op_table(self.lut_values, mult, bounds[0], bounds[1])
                return ops.lut_per_tensor(inputs, self.lut_values, thr_div, self._thr_mul0, mult, bounds[0], bounds[1],
                                          table, step, None, steps)
        return ops.lut_per_tensor(inputs, self.lut_values, thr_div, self._thr_mul0, mult, cmin, cmax,
                                  self._lut_table_torch, step, div64, self.__dict__.get("_lut_steps_torch"))
