"""One call -- or none -- between a model built on the reference's API and the launches this package can do for it.

The reference re-quantizes every wrapped weight inside its wrapper's forward, one call per weight per forward
(pytorch/quantize_wrapper.py:228-240), and that is what a model loaded with ``pytorch_load_quantized_model``
(pytorch/load_model.py:23-34) does here too unless something collects the work: ResNet-50's 54 weights are 54 launches,
each ~4 us of host time for ~1-6 us of GPU work.  ``accelerate(model)`` installs the pre-packed batched launch of
``pytorch/batching.py`` in its stand-aside form (``auto=True``): on a GPU the weights of ALL wrappers are re-quantized from
their current float values by ONE launch per storage type in front of the model's forward (same bits as the per-layer
calls); on the CPU, under a ``torch.jit`` trace or ``torch.compile``, or for anything the launch cannot take, the forward
is what it was.  ``pytorch_load_quantized_model`` and ``compat.load_reference_model`` call it on what they load unless
``MCTQ_AUTO_BATCH=0``.

What changes for the caller (the documented identity caveat of ``batch_weight_quantization(reuse_buffers=True)``): the
quantized weight a wrapper installs on its layer is the same tensor object on every forward, rewritten in place, where the
reference installs a fresh tensor; values are identical.  ``decelerate(model)`` removes the hook.

``accelerate(model, example_inputs=(x,))`` additionally captures the whole forward -- weight re-quantization, layers,
activation holders -- into one hipGraph (``pytorch/graphs.py``) and returns the replaying callable (inputs of another shape
run the eager forward); that returns static output buffers, so it is never done implicitly.
"""
from __future__ import annotations

import os
from typing import Optional, Sequence

import torch
import torch.nn as nn

_KEY = "_mctq_accelerated"          # the handle, kept in the root module's __dict__ (not a parameter, buffer or sub-module)


def auto_batch_enabled() -> bool:
    """``MCTQ_AUTO_BATCH`` (default on): loaded models get ``accelerate`` applied; "0" / "off" / "false" = the per-layer
    calls of the reference, exactly as before."""
    return os.environ.get("MCTQ_AUTO_BATCH", "1").strip().lower() not in ("0", "off", "false", "no", "")


def _has_wrapped_weights(model: nn.Module) -> bool:
    from mct_quantizers_amd.pytorch.containers import PytorchQuantizationWrapper
    return any(isinstance(m, PytorchQuantizationWrapper) and m.is_weights_quantization and m.is_str_attr
               for m in model.modules())


def accelerated(model: nn.Module):
    """The handle ``accelerate`` installed on ``model`` (``BatchedWeightQuantization``), or None."""
    return model.__dict__.get(_KEY)


def accelerate(model: nn.Module, example_inputs: Optional[Sequence[torch.Tensor]] = None):
    """Install the one-launch-per-forward weight re-quantization on ``model`` (idempotent) and return the model; with
    ``example_inputs`` capture the whole forward into one hipGraph as well and return the replaying callable."""
    if not isinstance(model, nn.Module):
        raise TypeError("accelerate() takes a torch.nn.Module")
    if example_inputs is not None:
        from mct_quantizers_amd.pytorch.graphs import capture_forward
        decelerate(model)                                  # the captured forward brings its own (non-auto) batcher
        return capture_forward(model, *example_inputs, strict=False)     # other shapes: the eager forward, not an error
    if model.__dict__.get(_KEY) is None and _has_wrapped_weights(model):
        from mct_quantizers_amd.pytorch.batching import batch_weight_quantization
        model.__dict__[_KEY] = batch_weight_quantization(model, reuse_buffers=True, auto=True)
    return model


def decelerate(model: nn.Module) -> nn.Module:
    """Remove what ``accelerate`` installed: every wrapper calls its own quantizer again."""
    handle = model.__dict__.pop(_KEY, None)
    if handle is not None:
        handle.remove()
    return model


def accelerate_loaded(obj):
    """What the loaders call on the object ``torch.load`` gave them."""
    if auto_batch_enabled() and isinstance(obj, nn.Module):
        accelerate(obj)
    return obj
