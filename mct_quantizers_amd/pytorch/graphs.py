"""Replay a wrapped model's forward from ONE hipGraph.

Small-batch inference of an MCT-exported model is launch-bound: every wrapped layer contributes a weight
re-quantization, the layer itself and an activation quantizer, each a few microseconds of GPU work behind 4-40 us of
host dispatch.  Every kernel of this package is legal under stream capture (no allocation, no synchronisation, no
device->host read; launch state is derived at construction), so the whole forward -- including the per-forward weight
re-quantization the reference's wrappers do (pytorch/quantize_wrapper.py:228-240) -- can be captured once and
replayed: ``profiles/r02/e2e_wrapped_linear_stack.log``: 36 wrapped convolutions at batch 1, 1.68 ms eager ->
0.33 ms replayed with the weights batched into one launch.  (ATen's ``fake_quantize_per_channel_affine`` cannot be
captured at all: its zero-point range check reads the device.)

The captured graph re-quantizes from the CURRENT float weights on every replay: in-place weight updates are seen,
exactly as in eager mode.  Shapes, dtypes and the module structure are frozen at capture time.
"""
from __future__ import annotations

from typing import Tuple

import torch
import torch.nn as nn

from mct_quantizers_amd.pytorch.batching import batch_weight_quantization


class CapturedForward:
    """``captured(x, ...)`` copies the inputs into the graph's static buffers, replays, and returns the static outputs
    (valid until the next call; ``.clone()`` what must outlive it)."""

    def __init__(self, model: nn.Module, example_inputs: Tuple[torch.Tensor, ...], batch_weights: bool, warmup: int):
        if not example_inputs or not all(isinstance(t, torch.Tensor) and t.is_cuda for t in example_inputs):
            raise TypeError("capture_forward takes GPU tensors as example inputs")
        self.model = model
        self._batcher = batch_weight_quantization(model, reuse_buffers=True) if batch_weights else None
        self._static_in = tuple(t.detach().clone() for t in example_inputs)
        was_training = model.training
        model.eval()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(max(1, warmup)):                 # allocator, lazy library state, the batcher's plan
                model(*self._static_in)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph), torch.no_grad():
            self._static_out = model(*self._static_in)
        model.train(was_training)

    def __call__(self, *inputs: torch.Tensor):
        if len(inputs) != len(self._static_in):
            raise TypeError(f"captured with {len(self._static_in)} inputs, called with {len(inputs)}")
        for dst, src in zip(self._static_in, inputs):
            if src.shape != dst.shape or src.dtype != dst.dtype:
                raise ValueError(f"captured for {tuple(dst.shape)} {dst.dtype}, got {tuple(src.shape)} {src.dtype}")
            dst.copy_(src)
        self.graph.replay()
        return self._static_out

    def release(self):
        """Drop the graph and restore per-layer weight quantization on the model."""
        if self._batcher is not None:
            self._batcher.remove()
            self._batcher = None
        self.graph = None


def capture_forward(model: nn.Module, *example_inputs: torch.Tensor, batch_weights: bool = True,
                    warmup: int = 3) -> CapturedForward:
    """Capture ``model(*example_inputs)`` (inference, no grad) into one hipGraph and return the replaying callable.
    ``batch_weights``: re-quantize all wrapped weights in one launch inside the graph (``pytorch/batching.py``)."""
    return CapturedForward(model, example_inputs, batch_weights, warmup)
