"""Re-quantize ALL wrapped layers' weights of a model in one launch per forward.

The reference quantizes each wrapped layer's weights inside that layer's forward
(pytorch/quantize_wrapper.py:228-240: ``for name, weight, quantizer in self._weights_vars: quantizer(weight)``),
i.e. one kernel launch per weight per forward -- tens per model, each paying its own host cost and ~2 us of GPU
ramp/drain (profiles/r02).  ``batch_weight_quantization(model)`` keeps those semantics (weights are re-quantized
on EVERY forward from the current float weights; nothing is cached across forwards) but moves the work of all
wrappers in front of the model's forward, where it is ONE call of ``ops.fq_batched`` -> ``mctq_fq_batched``
(include/mctq_hip.h).  Each wrapper then installs the tensor prepared for it instead of calling its quantizer.
Outputs are bit-identical to the per-layer calls (same AffineOp arithmetic; tested against the oracle).

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

    def __init__(self, model: nn.Module):
        self.model = model
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

    def quantize_now(self) -> int:
        """Quantize every participating weight in one batched launch and hand the results to the wrappers.
        Returns the number of tensors quantized."""
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
        for m in self.model.modules():
            if isinstance(m, PytorchQuantizationWrapper):
                m.__dict__.pop("_prequantized", None)


def batch_weight_quantization(model: nn.Module) -> BatchedWeightQuantization:
    """Install the batched weight re-quantization on ``model`` (a forward pre-hook on the given module)."""
    return BatchedWeightQuantization(model)
