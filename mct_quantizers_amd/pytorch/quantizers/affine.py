"""Affine (round -> clamp -> dequant) inferable quantizers: symmetric, power-of-two and uniform,
for weights (per-tensor / per-channel) and activations (per-tensor).

Public contract mirrored from the reference (constructor signatures, attributes, assertion
messages, reuse cache, side effects) -- see, relative to /root/reference/mct_quantizers/pytorch/quantizers/:
  base_pytorch_inferable_quantizer.py:24-62      flags + enable_* methods
  base_symmetric_inferable_quantizer.py:32-60    thresholds -> scales / clamp domain
  base_uniform_inferable_quantizer.py:33-66      ranges -> adjusted ranges
  weights_inferable_quantizers/weights_{symmetric,pot,uniform}_inferable_quantizer.py
  activation_inferable_quantizers/activation_{symmetric,pot,uniform}_inferable_quantizer.py

What differs is underneath ``__call__``: a GPU tensor goes through ONE fused gfx950 kernel
(mct_quantizers_amd/csrc/mctq_kernels.hip) instead of ATen's fake_quantize_*_cachemask kernels.
The ONNX-export branch of the reference (``_use_custom_impl and torch.jit.is_tracing()``) lives in
onnx_export.py; the quantizers below only switch into it.
"""
from typing import List

import numpy as np
import torch

from mct_quantizers_amd.common.registry import (BaseInferableQuantizer, QuantizationMethod, QuantizationTarget,
                                               QuantizerID, mark_quantizer)
from mct_quantizers_amd.hip import ops
from mct_quantizers_amd.pytorch.quantizer_utils import fix_range_to_include_zero, get_working_device


_is_compiling = torch.compiler.is_compiling


def _wide(q) -> bool:
    """The clamp domain does not fit the kernels' float32 bounds (num_bits > 24): ATen's operator runs instead."""
    lo, hi = q.__dict__.get("min_quantized_domain", 0), q.__dict__.get("max_quantized_domain", 0)
    try:
        return max(abs(int(lo)), abs(int(hi))) > (1 << _MAX_BITS)
    except (TypeError, ValueError):
        return True


class _PerTensorPlanMixin:
    """Pre-packed launch arguments of the per-tensor activation quantizers (compiled binding's AffinePlan).

    The plan is a function of a few public attributes (``_plan_attrs``); assigning to any of them drops it, so
    ``q.scale = 0.5`` takes effect on the next call exactly as in the reference, which reads the attributes on
    every call (activation_uniform_inferable_quantizer.py:124-128)."""
    _plan_attrs = frozenset()

    def __setattr__(self, name, value):
        object.__setattr__(self, name, value)
        if name in self._plan_attrs:
            self.__dict__["_plan"] = None

    def _plan_args(self):
        raise NotImplementedError

    def _make_plan(self):
        """AffinePlan, or False when the compiled binding is not in use (CPU-only process, MCTQ_BINDING=ctypes)."""
        fast = ops._fast_mod()
        plan = False
        if fast is not None and not _wide(self):
            try:
                plan = fast.AffinePlan(*self._plan_args())
            except (AttributeError, TypeError, ValueError):   # half-constructed object / non-numeric attribute
                plan = False
        self.__dict__["_plan"] = plan
        return plan

    def batch_item(self, inputs: torch.Tensor):
        """(x, scale[1], zero_point[1], None, qmin, qmax) for ``ops.fq_batched`` / ``BatchPlan``: this quantizer's call
        on ``inputs`` as one entry of a batched launch (a stream of activation batches quantized together).  The
        1-element device tensors hold float32(scale) -- the value the per-tensor launch narrows the Python double to --
        and are rebuilt whenever the public attributes change."""
        scale, zp, qmin, qmax = self._plan_args()
        key = (scale, zp, inputs.device)
        hit = self.__dict__.get("_batch_params")
        if hit is None or hit[0] != key:
            hit = (key, torch.tensor([scale], dtype=torch.float64).to(torch.float32).to(inputs.device),
                   torch.tensor([zp], dtype=torch.int32, device=inputs.device))
            self.__dict__["_batch_params"] = hit
        return inputs, hit[1], hit[2], None, qmin, qmax


_MAX_BITS = 24      # the kernels hold the clamp bounds in float32 (exact for |q| <= 2^24); wider domains -- which the
                    # reference accepts and hands to ATen as int64 bounds (base_symmetric_inferable_quantizer.py:53-60) --
                    # run ATen's own operator on the tensor's device (``_wide``): same results as the reference there


class BasePyTorchInferableQuantizer(BaseInferableQuantizer):
    """Base of all PyTorch inference-time quantizers: behaviour flags and the output-reuse cache."""

    def __init__(self):
        super().__init__()
        self._use_custom_impl = False
        # reuse: compute the quantized tensor once and hand the same object back afterwards
        self.reuse = False
        self.enable_reuse = False
        self.quantizer_first_run = True
        self.resue_outputs = None          # [sic] attribute name is part of the reference's surface

    def enable_custom_impl(self):
        self._use_custom_impl = True

    def enable_reuse_quantizer(self):
        self.enable_reuse = True
        self.quantizer_first_run = True

    def disable_reuse_quantizer(self):
        self.enable_reuse = False

    def enable_versioned_reuse(self):
        """Extension (not in the reference): keep the quantized tensor until the input changes.

        ``enable_reuse_quantizer`` returns the first result forever; this variant re-quantizes whenever the
        input is a different tensor or has been modified in place (torch bumps ``Tensor._version`` on every
        in-place write, e.g. an optimizer step or ``copy_``), so inference loops stop paying for the
        re-quantization of unchanged weights without any risk of serving stale ones."""
        self.__dict__["_versioned_reuse"] = True
        self.__dict__["_versioned_key"] = None

    def disable_versioned_reuse(self):
        self.__dict__["_versioned_reuse"] = False
        self.__dict__["_versioned_key"] = None

    def __call__(self, inputs: torch.Tensor):
        raise NotImplementedError(f'{self.__class__.__name__} did not implement __call__')  # pragma: no cover

    # -- pickles: objects saved by the reference (or by an older build) carry only the public attributes;
    #    the private launch state (host copies of scalars, decision tables) is rebuilt on load.
    def __getstate__(self):
        state = dict(self.__dict__)
        for k in ("_plan", "_plan_key", "_versioned_key", "_versioned_pending", "_batch_params"):   # binding handles, weak references
            state.pop(k, None)
        return state

    def __setstate__(self, state):
        self.__dict__.update(state)
        self.__dict__.pop("_plan", None)
        self.__dict__.pop("_plan_key", None)
        self._rebuild_launch_state()

    def _rebuild_launch_state(self):
        """Recreate private, derived attributes from the public ones (idempotent)."""

    @staticmethod
    def _to_working_device(t):
        return t.to(get_working_device()) if isinstance(t, torch.Tensor) else t

    # -- reuse cache helpers shared by the weights quantizers (weights_symmetric...py:128-129,153-155)
    def _cached(self, inputs=None):
        if self.enable_reuse and not self.quantizer_first_run:
            return True
        if inputs is not None and self.__dict__.get("_versioned_reuse") and isinstance(inputs, torch.Tensor):
            # the tensor OBJECT (weak reference: a freed tensor's address can be handed to another one), its version
            # counter and its storage address (``.data`` of a Parameter can be re-pointed without a version bump)
            import weakref
            key = (inputs.data_ptr(), inputs._version, tuple(inputs.shape), inputs.stride(), inputs.dtype,
                   inputs.device)
            old = self.__dict__.get("_versioned_key")
            hit = (old is not None and old[1] == key and old[0]() is inputs and self.resue_outputs is not None)
            try:
                self.__dict__["_versioned_pending"] = (weakref.ref(inputs), key)
            except TypeError:
                self.__dict__["_versioned_pending"] = None
            return hit
        return False

    def _remember(self, outputs):
        if self.enable_reuse and self.quantizer_first_run:
            self.resue_outputs = outputs
            self.quantizer_first_run = False
        elif self.__dict__.get("_versioned_reuse"):
            self.resue_outputs = outputs
            self.__dict__["_versioned_key"] = self.__dict__.get("_versioned_pending")
        return outputs


def _is_pot(values: np.ndarray) -> bool:
    lg = np.log2(values.flatten())
    return bool(np.all(np.round(lg) == lg))


@mark_quantizer(quantization_target=None,
                quantization_method=[QuantizationMethod.SYMMETRIC],
                identifier=QuantizerID.INFERABLE)
class BaseSymmetricInferableQuantizer(BasePyTorchInferableQuantizer):

    def __init__(self, num_bits: int, threshold: List[float], signed: bool):
        super().__init__()
        assert isinstance(threshold, list), f'Threshold is expected to be a list, but is of type {type(threshold)}'
        self.signed = signed
        self.threshold_np = np.asarray(threshold)
        self.num_bits = num_bits
        levels = 2 ** (num_bits - 1) if signed else 2 ** num_bits
        self.min_quantized_domain = -levels if signed else 0
        self.max_quantized_domain = levels - 1
        self.scales = self.threshold_np / levels          # float64 here; subclasses narrow it


@mark_quantizer(quantization_target=None,
                quantization_method=[QuantizationMethod.UNIFORM],
                identifier=QuantizerID.INFERABLE)
class BaseUniformInferableQuantizer(BasePyTorchInferableQuantizer):

    def __init__(self, num_bits: int, min_range: List[float], max_range: List[float]):
        super().__init__()
        assert isinstance(min_range, list), f'min_range is expected to be a list, but is of type {type(min_range)}'
        assert isinstance(max_range, list), f'max_range is expected to be a list, but is of type {type(max_range)}'
        for _min, _max in zip(min_range, max_range):
            assert _min < _max, f"Max range must be greater than min value but min is {_min} and max is {_max}"

        # Parameter math runs in float32 on the HOST and only the results move to the working device:
        # ATen's GPU kernels evaluate tensor / python_scalar as tensor * (1 / scalar), which can differ
        # from the CPU result by one ulp, and the parameters must not depend on the machine.
        lo = torch.from_numpy(np.asarray(min_range).astype(np.float32))
        hi = torch.from_numpy(np.asarray(max_range).astype(np.float32))
        lo, hi = fix_range_to_include_zero(lo, hi, num_bits)
        self._min_range_host, self._max_range_host = lo, hi
        dev = get_working_device()
        self.min_range, self.max_range = lo.to(dev), hi.to(dev)
        self.num_bits = num_bits
        self.min_quantized_domain = 0
        self.max_quantized_domain = 2 ** num_bits - 1


class _WeightsAffineMixin:
    """Per-tensor / per-channel dispatch shared by the symmetric and the uniform weights quantizers.

    Expects: self.scales (float32 device tensor [C]), self.zero_points (int32 device tensor [C]),
    self.per_channel, self.channel_axis, the clamp domain.  The launch state derived from them -- flattened
    parameter views, host copies of the single scale / zero point for the per-tensor launch (no device->host read
    per call), the "all zero points are zero" flag and the compiled binding's pre-packed AffinePlan -- is keyed on
    the identity AND the version counter of the two public tensors, so replacing them or editing them in place
    (``q.zero_points[0] = 5``) takes effect on the next call, as it does in the reference, which reads the
    attributes on every call (weights_symmetric_inferable_quantizer.py:139-151).
    """

    def _rebuild_launch_state(self):
        self.scales = self._to_working_device(self.scales)
        self.zero_points = self._to_working_device(self.zero_points)
        self.__dict__.pop("_plan_key", None)
        if hasattr(self, "per_channel"):
            self._refresh()                # now, not at the first call: that call may sit inside hipGraph capture

    def _launch_key(self):
        s, z = self.scales, self.zero_points
        return (s, z, s._version, z._version, self.per_channel, self.channel_axis, self.min_quantized_domain,
                self.max_quantized_domain)

    def _refresh(self, host_scales=None, host_zps=None):
        """(Re)derive the launch state from the public attributes.  Constructors pass the HOST copies of the
        parameters they just computed; otherwise (public attributes replaced or edited in place, objects
        un-pickled) the values are read back from the device -- a synchronising read that is not legal
        inside hipGraph capture, so change quantizer attributes outside captured regions."""
        d = self.__dict__
        s, z = self.scales, self.zero_points
        d["_scales_flat"] = s.flatten().contiguous()
        d["_zps_flat"] = z.flatten().contiguous()
        if host_scales is None or host_zps is None:
            host_scales = d["_scales_flat"][:1].cpu()
            zps_nonzero = bool(torch.any(d["_zps_flat"] != 0).item())
            host_zp0 = int(d["_zps_flat"][0].item()) if d["_zps_flat"].numel() else 0
        else:
            host_scales, host_zps = host_scales.reshape(-1), host_zps.reshape(-1)
            zps_nonzero = bool(torch.any(host_zps != 0))
            host_zp0 = int(host_zps[0]) if host_zps.numel() else 0
        # ATen validates the zero points of the per-channel operator on EVERY call (a device -> host read, most of its
        # 46 us host cost); here once per parameter change (the call raises ATen's message, _quantize_weights)
        zmin, zmax = ((int(host_zps.min()), int(host_zps.max())) if host_zps is not None and host_zps.numel()
                      else (int(d["_zps_flat"].min().item()), int(d["_zps_flat"].max().item())) if d["_zps_flat"].numel() else (0, 0))
        d["_zp_out_of_range"] = bool(self.per_channel) and (zmin < self.min_quantized_domain or zmax > self.max_quantized_domain)
        d["_zps_all_zero"] = not zps_nonzero                                     # symmetric: skip the table
        d["_scale0"] = float(host_scales[0]) if host_scales.numel() else 1.0
        d["_zp0"] = host_zp0
        plan = None
        fast = ops._fast_mod() if d["_scales_flat"].is_cuda else None
        if fast is not None and not _wide(self):
            if self.per_channel:
                plan = fast.AffinePlan(d["_scales_flat"], None if d["_zps_all_zero"] else d["_zps_flat"],
                                       int(self.channel_axis), self.min_quantized_domain, self.max_quantized_domain)
            else:
                plan = fast.AffinePlan(d["_scale0"], d["_zp0"], self.min_quantized_domain, self.max_quantized_domain)
        d["_plan"] = plan
        d["_plan_key"] = self._launch_key()

    def _current(self):
        key = self.__dict__.get("_plan_key")
        if key is None:
            self._refresh()
            return
        s, z = self.scales, self.zero_points
        if (key[0] is not s or key[1] is not z or key[2] != s._version or key[3] != z._version
                or key[4] != self.per_channel or key[5] != self.channel_axis
                or key[6] != self.min_quantized_domain or key[7] != self.max_quantized_domain):
            self._refresh()

    def quantize_to_codes(self, inputs: torch.Tensor, packed4: bool = False):
        """Extension (not in the reference): the integer clamp indices as int8/uint8 plus the parameters that
        dequantize them, ``(codes - zero_points) * scales`` == ``self(inputs)`` bit for bit.
        ``packed4`` (num_bits <= 4): two codes per byte (``ops.unpack4`` undoes it).
        Returns (codes, scales float32 [C or 1], zero_points int32 [C or 1])."""
        self._current()
        axis = self.channel_axis if self.per_channel else None
        codes = ops.fq_codes(inputs, self._scales_flat, self._zps_flat, axis, self.min_quantized_domain,
                             self.max_quantized_domain, self._scale0, self._zp0, packed4)
        return codes, self._scales_flat, self._zps_flat

    def batch_item(self, inputs: torch.Tensor):
        """(x, scales, zero_points | None, axis | None, qmin, qmax) for ``ops.fq_batched``: this quantizer's call on
        ``inputs`` as one entry of a batched launch."""
        self._current()
        d = self.__dict__
        return (inputs, d["_scales_flat"], None if d["_zps_all_zero"] else d["_zps_flat"],
                int(self.channel_axis) if self.per_channel else None, self.min_quantized_domain,
                self.max_quantized_domain)

    def _quantize_weights(self, inputs: torch.Tensor) -> torch.Tensor:
        if inputs.requires_grad:                # the reference flips this on the caller's tensor (read first: a tensor
            inputs.requires_grad = False        # that already has it off stays traceable by dynamo)
        if _is_compiling():
            # torch.compile: the graph records the library ops on the public attributes, as the reference's call sites
            # read them (weights_symmetric_inferable_quantizer.py:139-151); no launch-state bookkeeping in a graph
            if self.per_channel:
                return ops.fq_per_channel(inputs, self.scales.flatten(), self.zero_points.flatten(), self.channel_axis,
                                          self.min_quantized_domain, self.max_quantized_domain)
            return ops.fq_per_tensor_tqp(inputs, self.scales, self.zero_points, self.min_quantized_domain,
                                         self.max_quantized_domain)
        self._current()
        d = self.__dict__
        if d["_zp_out_of_range"]:
            raise RuntimeError("`zero_point` must be between `quant_min` and `quant_max`.")
        plan = d["_plan"]
        if plan is not None and not _is_compiling():
            y = plan(inputs)                    # compiled binding: checks + allocation + launch in one call
            if y is not NotImplemented:
                return y
        if self.per_channel:
            return ops.fq_per_channel(inputs, d["_scales_flat"], d["_zps_flat"], self.channel_axis,
                                      self.min_quantized_domain, self.max_quantized_domain, d["_zps_all_zero"])
        return ops.fq_per_tensor(inputs, d["_scale0"], d["_zp0"],
                                 self.min_quantized_domain, self.max_quantized_domain)


@mark_quantizer(quantization_target=QuantizationTarget.Weights,
                quantization_method=[QuantizationMethod.SYMMETRIC],
                identifier=QuantizerID.INFERABLE)
class WeightsSymmetricInferableQuantizer(_WeightsAffineMixin, BaseSymmetricInferableQuantizer):
    """Symmetric signed weights quantizer, per tensor or per channel."""

    def __init__(self, num_bits: int, threshold: List[float], per_channel: bool, channel_axis: int = None):
        super().__init__(num_bits=num_bits, threshold=threshold, signed=True)
        if per_channel:
            assert channel_axis is not None, 'Channel axis is missing in per channel quantization'
            assert len(threshold) >= 1, \
                f'In per-channel quantization threshold should be of length >= 1 but is {len(threshold)}'
        else:
            assert len(threshold) == 1, \
                f'In per-tensor quantization threshold should be of length 1 but is {len(threshold)}'
        self.per_channel = per_channel
        self.channel_axis = channel_axis

        dev = get_working_device()
        host_scales = torch.from_numpy(self.scales.astype(np.float32))
        host_zps = torch.zeros(len(threshold), dtype=torch.int32)
        self.scales = host_scales.to(dev)
        self.zero_points = host_zps.to(dev)
        self._refresh(host_scales, host_zps)

    _export_function = "WeightsSymmetricF"

    def __call__(self, inputs: torch.Tensor) -> torch.Tensor:
        if self._cached(inputs):
            return self.resue_outputs
        if self._use_custom_impl and torch.jit.is_tracing():
            from mct_quantizers_amd.pytorch.quantizers import onnx_export
            return self._remember(getattr(onnx_export, self._export_function).apply(
                inputs, self.num_bits, self.threshold_np, self.per_channel, self.channel_axis))
        return self._remember(self._quantize_weights(inputs))


@mark_quantizer(quantization_target=QuantizationTarget.Weights,
                quantization_method=[QuantizationMethod.POWER_OF_TWO],
                identifier=QuantizerID.INFERABLE)
class WeightsPOTInferableQuantizer(WeightsSymmetricInferableQuantizer):
    """Symmetric weights quantizer whose thresholds must be powers of two."""
    _export_function = "WeightsPOTF"

    def __init__(self, num_bits: int, threshold: List[float], per_channel: bool, channel_axis: int = None):
        super().__init__(num_bits=num_bits, threshold=threshold, per_channel=per_channel, channel_axis=channel_axis)
        self.threshold = threshold
        assert _is_pot(self.threshold_np), f'Expected threshold to be power of 2 but is {threshold}'


@mark_quantizer(quantization_target=QuantizationTarget.Weights,
                quantization_method=[QuantizationMethod.UNIFORM],
                identifier=QuantizerID.INFERABLE)
class WeightsUniformInferableQuantizer(_WeightsAffineMixin, BaseUniformInferableQuantizer):
    """Unsigned uniform weights quantizer over [min_range, max_range] (adjusted to contain 0)."""

    def __init__(self, num_bits: int, min_range: List[float], max_range: List[float], per_channel: bool,
                 channel_axis: int = None):
        super().__init__(num_bits=num_bits, min_range=min_range, max_range=max_range)
        if per_channel:
            assert channel_axis is not None, 'Channel axis is missing in per channel quantization'
            assert len(min_range) >= 1, \
                f'In per-channel quantization min_range should be of length >= 1 but is {len(min_range)}'
            assert len(max_range) >= 1, \
                f'In per-channel quantization max_range should be of length >= 1 but is {len(max_range)}'
        else:
            assert len(min_range) == 1, \
                f'In per-tensor quantization min_range should be of length 1 but is {len(min_range)}'
            assert len(max_range) == 1, \
                f'In per-tensor quantization max_range should be of length 1 but is {len(max_range)}'
        self.per_channel = per_channel
        self.channel_axis = channel_axis

        lo, hi = self._min_range_host, self._max_range_host
        self.adjusted_min_range_np = lo.numpy()
        self.adjusted_max_range_np = hi.numpy()

        # step of the grid, and the (positive) zero point: TRUNCATION of min/scale, negated (host float32)
        scales = (hi - lo) / (2 ** num_bits - 1)
        zero_points = -(lo / scales).int()
        dev = get_working_device()
        self.scales = scales.to(dev)
        self.zero_points = zero_points.to(dev)
        self._refresh(scales, zero_points)

    def __call__(self, inputs: torch.Tensor) -> torch.Tensor:
        if self._cached(inputs):
            return self.resue_outputs
        if self._use_custom_impl and torch.jit.is_tracing():
            from mct_quantizers_amd.pytorch.quantizers.onnx_export import WeightsUniformF
            return self._remember(WeightsUniformF.apply(inputs, self.num_bits, self.adjusted_min_range_np,
                                                        self.adjusted_max_range_np, self.per_channel,
                                                        self.channel_axis))
        return self._remember(self._quantize_weights(inputs))


@mark_quantizer(quantization_target=QuantizationTarget.Activation,
                quantization_method=[QuantizationMethod.SYMMETRIC],
                identifier=QuantizerID.INFERABLE)
class ActivationSymmetricInferableQuantizer(_PerTensorPlanMixin, BaseSymmetricInferableQuantizer):
    """Symmetric activation quantizer (per tensor only), signed or unsigned."""
    _plan_attrs = frozenset(("scales", "zero_points", "min_quantized_domain", "max_quantized_domain"))

    def _plan_args(self):
        return float(self.scales), int(self.zero_points), self.min_quantized_domain, self.max_quantized_domain

    def _rebuild_launch_state(self):
        self.__dict__["_plan"] = None

    def __init__(self, num_bits: int, threshold: List[float], signed: bool):
        super().__init__(num_bits=num_bits, threshold=threshold, signed=signed)
        assert len(threshold) == 1, ('For activation, only per-tensor quantization is supported. Thus, threshold '
                                     f'should be of length 1 but is {len(threshold)}')
        assert self.threshold_np.shape[0] == 1
        self.threshold_np = self.threshold_np[0]
        assert len(self.scales) == 1, ('For activation, quantization per channel is not supported and threshold '
                                       f'should be of length 1 but is {len(threshold)}')
        self.scales = float(self.scales[0])      # stays a Python double; narrowed to float32 at launch
        self.zero_points = 0
        self._make_plan()

    def quantize_to_codes(self, inputs: torch.Tensor, packed4: bool = False):
        """Extension: (codes int8/uint8, scale float, zero_point int) with (codes - zero_point) * float32(scale)
        == self(inputs); ``packed4``: two 4-bit codes per byte."""
        codes = ops.fq_codes(inputs, None, None, None, self.min_quantized_domain, self.max_quantized_domain,
                             self.scales, self.zero_points, packed4)
        return codes, self.scales, self.zero_points

    _export_function = "ActivationSymF"

    def __call__(self, inputs: torch.Tensor):
        # Small activations are launch-bound: a plain eager GPU tensor goes from here to the kernel launch in ONE
        # call of the compiled binding (it returns NotImplemented for everything else: CPU tensors, fx proxies,
        # tensor subclasses, an active torch.jit trace -- those take the general route below).  The launch records
        # nothing for autograd, so the reference's no_grad context is only needed on the general route.
        plan = self.__dict__.get("_plan")
        if plan is None:
            plan = self._make_plan()
        if plan is not False and not _is_compiling():     # torch.compile must see torch.ops.mctq_amd.* instead
            y = plan(inputs)
            if y is not NotImplemented:
                return y
        if self._use_custom_impl and torch.jit.is_tracing():
            from mct_quantizers_amd.pytorch.quantizers import onnx_export
            return getattr(onnx_export, self._export_function).apply(inputs, self.threshold_np, self.signed,
                                                                     self.num_bits)
        with torch.no_grad():
            return ops.fq_per_tensor(inputs, self.scales, self.zero_points,
                                     self.min_quantized_domain, self.max_quantized_domain)


@mark_quantizer(quantization_target=QuantizationTarget.Activation,
                quantization_method=[QuantizationMethod.POWER_OF_TWO],
                identifier=QuantizerID.INFERABLE)
class ActivationPOTInferableQuantizer(ActivationSymmetricInferableQuantizer):
    """Symmetric activation quantizer whose threshold must be a power of two."""
    _export_function = "ActivationPOTF"

    def __init__(self, num_bits: int, threshold: List[float], signed: bool):
        super().__init__(num_bits=num_bits, signed=signed, threshold=threshold)
        assert _is_pot(self.threshold_np), f'Expected threshold to be power of 2 but is {threshold}'


@mark_quantizer(quantization_target=QuantizationTarget.Activation,
                quantization_method=[QuantizationMethod.UNIFORM],
                identifier=QuantizerID.INFERABLE)
class ActivationUniformInferableQuantizer(_PerTensorPlanMixin, BaseUniformInferableQuantizer):
    """Unsigned uniform activation quantizer (per tensor only)."""
    _plan_attrs = frozenset(("scale", "zero_point", "min_quantized_domain", "max_quantized_domain"))

    def _plan_args(self):
        return float(self.scale), int(self.zero_point), self.min_quantized_domain, self.max_quantized_domain

    def _rebuild_launch_state(self):
        self.__dict__["_plan"] = None

    def __init__(self, num_bits: int, min_range: List[float], max_range: List[float]):
        super().__init__(num_bits=num_bits, min_range=min_range, max_range=max_range)
        assert len(min_range) == 1, ('For activation, only per-tensor quantization is supported. Thus, min_range '
                                     f'should be of length 1 but is {len(min_range)}')
        assert len(max_range) == 1, ('For activation, only per-tensor quantization is supported. Thus, max_range '
                                     f'should be of length 1 but is {len(max_range)}')
        self.min_range = self._min_range_host[0].item()
        self.max_range = self._max_range_host[0].item()
        self.scale = float((self.max_range - self.min_range) / ((2 ** num_bits) - 1))
        self.zero_point = int(-np.round(self.min_range / self.scale))   # round half even, in double
        self._make_plan()

    def quantize_to_codes(self, inputs: torch.Tensor, packed4: bool = False):
        """Extension: (codes uint8, scale float, zero_point int) with (codes - zero_point) * float32(scale)
        == self(inputs); ``packed4``: two 4-bit codes per byte."""
        codes = ops.fq_codes(inputs, None, None, None, self.min_quantized_domain, self.max_quantized_domain,
                             self.scale, self.zero_point, packed4)
        return codes, self.scale, self.zero_point

    def __call__(self, inputs: torch.Tensor):
        plan = self.__dict__.get("_plan")           # see ActivationSymmetricInferableQuantizer.__call__
        if plan is None:
            plan = self._make_plan()
        if plan is not False and not _is_compiling():
            y = plan(inputs)
            if y is not NotImplemented:
                return y
        if self._use_custom_impl and torch.jit.is_tracing():
            from mct_quantizers_amd.pytorch.quantizers.onnx_export import ActivationUniformF
            return ActivationUniformF.apply(inputs, self.min_range, self.max_range, self.num_bits)
        with torch.no_grad():
            return ops.fq_per_tensor(inputs, self.scale, self.zero_point,
                                     self.min_quantized_domain, self.max_quantized_domain)
