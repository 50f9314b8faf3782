"""Re-quantize ALL wrapped layers' weights of a model in one launch per forward.

The reference quantizes each wrapped layer's weights inside that layer's forward
(pytorch/quantize_wrapper.py:228-240: ``for name, weight, quantizer in self._weights_vars: quantizer(weight)``),
i.e. one kernel launch per weight per forward -- tens per model, each paying its own host cost and ~2 us of GPU
ramp/drain (profiles/r02).  ``batch_weight_quantization(model)`` keeps those semantics (weights are re-quantized
on EVERY forward from the current float weights; nothing is cached across forwards) but moves the work of all
wrappers in front of the model's forward, where it is ONE call of ``ops.fq_batched`` -> ``mctq_fq_batched``
(include/mctq_hip.h).  Each wrapper then installs the tensor prepared for it instead of calling its quantizer.
Outputs are bit-identical to the per-layer calls (same AffineOp arithmetic; tested against the oracle).

``reuse_buffers=True`` (opt-in) additionally keeps ONE persistent output tensor per weight and a pre-packed launch
(the compiled binding's ``BatchPlan``): a forward then costs one C call -- no allocation, no per-tensor Python --
and rewrites those tensors in place.  The values are the same; what changes is object identity: the quantized weight a
wrapper installs is the same tensor object every forward (the reference returns a fresh one), so do not hold on to a
previous forward's quantized weights across forwards in that mode.  The plan is built at the first forward from the
wrappers and quantizer parameters found then; in-place updates of the WEIGHTS are followed (that is the point), but
after replacing a quantizer's ``scales`` / ``zero_points`` or adding / removing wrappers call ``handle.refresh()``.
A wrapper uses its persistent tensor only in the forward whose pre-hook has just filled it (generation counter):
calling a sub-module directly, past the hook, falls back to that wrapper's own quantizer call.  54 ResNet-50 weights: 70 us -> host ~4 us + ~35 us
of GPU time (profiles/r02/requant_model_weights_batched.log).

Only the affine weights quantizers (symmetric / power-of-two / uniform, per tensor or per channel) take part;
LUT quantizers, quantizers with the reuse cache enabled, trainable quantizers and wrappers of positional
(functional) weights keep calling their quantizer as before.
"""
from __future__ import annotations

from typing import List, Tuple

import torch
import torch.nn as nn

from mct_quantizers_amd.hip import ops
from mct_quantizers_amd.pytorch.containers import PytorchQuantizationWrapper


class BatchedWeightQuantization:
    """Handle returned by ``batch_weight_quantization``; ``remove()`` restores per-layer quantization."""

    def __init__(self, model: nn.Module, reuse_buffers: bool = False):
        self.model = model
        self.reuse_buffers = reuse_buffers
        self._plan = None                 # (BatchPlan, generation cell, tensor count) in reuse_buffers mode
        self._hook = model.register_forward_pre_hook(self._before_forward)

    def _entries(self) -> List[Tuple[PytorchQuantizationWrapper, str, torch.Tensor, object]]:
        out = []
        for m in self.model.modules():
            if isinstance(m, PytorchQuantizationWrapper) and m.is_weights_quantization:
                for name, weight, quantizer in m.get_weights_vars():
                    if (hasattr(quantizer, "batch_item") and not quantizer.enable_reuse
                            and not quantizer.__dict__.get("_versioned_reuse")
                            and not (quantizer._use_custom_impl and torch.jit.is_tracing())):
                        out.append((m, name, weight, quantizer))
        return out

    def _build_plan(self, entries):
        fast = ops._fast_mod()
        if fast is None or not all(w.is_cuda for _, _, w, _ in entries):
            return None
        items, per_wrapper = [], {}
        for wrapper, name, weight, quantizer in entries:
            weight.requires_grad = False            # the side effect of the reference's weights quantizers
            x, scales, zps, axis, qmin, qmax = quantizer.batch_item(weight)
            if zps is None and axis is None and weight.dtype == torch.float64:
                zps = torch.zeros(1, dtype=torch.int32, device=weight.device)
            if not ops._is_dense(x.detach()):
                return None
            y = torch.empty_like(x.detach())
            items.append((x, y, scales, zps, axis, qmin, qmax))
            per_wrapper.setdefault(wrapper, {})[name] = y
        try:
            plan = fast.BatchPlan(items)
        except TypeError:
            return None                   # something the pre-packed launch cannot take: per-forward batching instead
        cell = [0]                        # generation: bumped once per forward, after the launch
        for wrapper, outs in per_wrapper.items():
            wrapper.__dict__["_prequantized_plan"] = (cell, outs)
            wrapper.__dict__.pop("_prequantized_seen", None)
        return plan, cell, len(entries)

    def refresh(self):
        """Rebuild the pre-packed plan at the next forward (after changing quantizer parameters or the model)."""
        self._drop_plan()

    def _drop_plan(self):
        self._plan = None
        for m in self.model.modules():
            if isinstance(m, PytorchQuantizationWrapper):
                m.__dict__.pop("_prequantized_plan", None)
                m.__dict__.pop("_prequantized_seen", None)

    def quantize_now(self) -> int:
        """Quantize every participating weight in one batched launch and hand the results to the wrappers.
        Returns the number of tensors quantized."""
        if self.reuse_buffers and not torch.jit.is_tracing():
            plan = self._plan
            if plan is None:
                entries = self._entries()
                plan = self._plan = self._build_plan(entries) if entries else None
            if plan is not None and plan[0]() is None:          # ONE C call: pointers re-read, one launch per 32 tensors
                plan[1][0] += 1
                return plan[2]
            self._drop_plan()
        entries = self._entries()
        if not entries:
            return 0
        items = []
        for _, _, weight, quantizer in entries:
            weight.requires_grad = False            # the side effect of the reference's weights quantizers
            items.append(quantizer.batch_item(weight))
        outs = ops.fq_batched(items)
        for (wrapper, name, _, _), y in zip(entries, outs):
            wrapper.__dict__.setdefault("_prequantized", {})[name] = y
        return len(entries)

    def _before_forward(self, module, args):
        self.quantize_now()
        return None

    def remove(self):
        self._hook.remove()
        self._drop_plan()
        for m in self.model.modules():
            if isinstance(m, PytorchQuantizationWrapper):
                m.__dict__.pop("_prequantized", None)


def batch_weight_quantization(model: nn.Module, reuse_buffers: bool = False) -> BatchedWeightQuantization:
    """Install the batched weight re-quantization on ``model`` (a forward pre-hook on the given module).
    ``reuse_buffers``: see the module docstring (persistent output tensors + pre-packed launch)."""
    return BatchedWeightQuantization(model, reuse_buffers)
