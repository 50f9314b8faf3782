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

import contextlib
import gc
from typing import Callable, List, Sequence, Tuple

import torch
import torch.nn as nn

from mct_quantizers_amd.hip import ops
from mct_quantizers_amd.pytorch.batching import batch_weight_quantization


@contextlib.contextmanager
def no_gc_while_capturing():
    """Stream capture forbids most runtime calls, and the cyclic garbage collector may run at any allocation: if it then
    frees an object whose destructor makes such a call -- an older hipGraph with its memory pool, dropped together with a
    model it formed a reference cycle with -- the process aborts (seen in a fuzz that builds and drops captured models:
    "Fatal Python error: Aborted ... Garbage-collecting" inside ``torch.cuda.graph``).  ``torch.cuda.graph`` no longer
    collects before it begins (``torch.compiler.config.force_cudagraph_gc`` is off by default since torch 2.x), so dead
    cycles can still be around; this keeps the collector off until the capture has ended -- they are freed right after."""
    was = gc.isenabled()
    gc.disable()
    try:
        yield
    finally:
        if was:
            gc.enable()


class CapturedForward:
    """``captured(x, ...)`` copies the inputs into the graph's static buffers, replays, and returns the static outputs
    (valid until the next call; ``.clone()`` what must outlive it)."""

    def __init__(self, model: nn.Module, example_inputs: Tuple[torch.Tensor, ...], batch_weights: bool, warmup: int,
                 strict: bool = True):
        self.strict = strict
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
        with no_gc_while_capturing(), torch.cuda.graph(self.graph), torch.no_grad():
            self._static_out = model(*self._static_in)
        model.train(was_training)

    def fits(self, inputs) -> bool:
        return (self.graph is not None and len(inputs) == len(self._static_in)
                and all(isinstance(src, torch.Tensor) and src.shape == dst.shape and src.dtype == dst.dtype
                        and src.device == dst.device for dst, src in zip(self._static_in, inputs)))

    def __call__(self, *inputs: torch.Tensor):
        if not self.fits(inputs):
            if self.strict:
                if len(inputs) != len(self._static_in):
                    raise TypeError(f"captured with {len(self._static_in)} inputs, called with {len(inputs)}")
                for dst, src in zip(self._static_in, inputs):
                    if src.shape != dst.shape or src.dtype != dst.dtype:
                        raise ValueError(f"captured for {tuple(dst.shape)} {dst.dtype}, got {tuple(src.shape)} {src.dtype}")
            # another shape / dtype / device (or a released graph): the eager forward, batched weights included
            with torch.no_grad():
                return self.model(*inputs)
        for dst, src in zip(self._static_in, inputs):
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
                    warmup: int = 3, strict: bool = True) -> CapturedForward:
    """Capture ``model(*example_inputs)`` (inference, no grad) into one hipGraph and return the replaying callable.
    ``batch_weights``: re-quantize all wrapped weights in one launch inside the graph (``pytorch/batching.py``).
    ``strict=False``: inputs of another shape / dtype / device run the eager forward instead of raising."""
    return CapturedForward(model, example_inputs, batch_weights, warmup, strict)


class CapturedStream:
    """A fixed-shape STREAM of activation batches through one quantizer / holder, ``depth`` batches per replay.

    Small activations are launch-bound: an eager ``holder(x)`` costs ~4-5 us of host time for ~1-2 us of GPU work
    (BASELINE config 3 at N = 1 / 8; reference call site pytorch/activation_quantization_holder.py:43-53).  The stream owns
    ``depth`` static input tensors; the producer writes batch i into ``stream.inputs[i]`` (or hands ``depth`` tensors to
    ``stream(batches)``, which copies them in), ``stream.run()`` quantizes all of them, and ``stream.outputs[i]`` holds
    batch i's result until the next replay.  Two ways to run the ``depth`` calls, same bits as the eager calls either way:

      fused   (affine activation quantizers: symmetric / power-of-two / uniform; LUT quantizers whose codebook has a
              decision table)  ONE launch of the batched kernel over all ``depth`` batches (``BatchPlan`` ->
              mctq_fq_batch_run / mctq_lutt_batch_run, per-tensor items): no hipGraph at all, one host call and one
              ramp / drain per ``depth`` batches -- config 3 at N = 1: 5.2 -> ~0.3 us per batch.
      graph   (anything else: wide codebooks, arbitrary callables)  one hipGraph of ``depth`` kernel nodes on one branch.
              Measured (profiles/r03/stream_probe.log): a replay costs ~12 us of fixed overhead, so a ONE-node graph is
              2-3x SLOWER than the eager call (14.5 vs 5.2 us) and pays only from depth >= 8 (2.2 us per batch at depth
              16); parallel branches (``lanes`` > 1) made small batches slower, so the default is 1.

    Anything that does not fit the captured shape / dtype / count falls back to eager calls."""

    def __init__(self, fn: Callable[[torch.Tensor], torch.Tensor], example: torch.Tensor, depth: int = 16, lanes: int = 1,
                 warmup: int = 2, mode: str = "auto"):
        if not isinstance(example, torch.Tensor) or not example.is_cuda:
            raise TypeError("capture_stream takes a GPU tensor as the example batch")
        if depth < 1 or lanes < 1:
            raise ValueError("depth and lanes must be >= 1")
        if mode not in ("auto", "fused", "graph"):
            raise ValueError("mode: auto | fused | graph")
        self.fn = fn
        self.depth, self.lanes = depth, min(lanes, depth)
        self.inputs: List[torch.Tensor] = [example.detach().clone() for _ in range(depth)]
        self.outputs: List[torch.Tensor] = [None] * depth
        self.graph, self._plan = None, None
        quantizer = getattr(fn, "activation_holder_quantizer", fn)
        bypass = bool(getattr(fn, "quantization_bypass", False))
        affine = hasattr(quantizer, "batch_item") and hasattr(quantizer, "_plan_args")
        lut = (not affine and hasattr(quantizer, "batch_item_lut") and example.is_contiguous()
               and quantizer.batch_item_lut(self.inputs[0]) is not None)
        fusable = (mode != "graph" and not bypass and (affine or lut) and ops._fast_mod() is not None
                   and ops._is_dense(example))
        if mode == "fused" and not fusable:
            raise TypeError("fused streams need an affine or decision-table LUT activation quantizer (or its holder) and "
                            "the compiled binding")
        if fusable:
            self.mode = "fused"
            items = []
            if affine:
                self.outputs = [torch.empty_like(x) for x in self.inputs]
                for x, y in zip(self.inputs, self.outputs):
                    xi, scale, zp, axis, qmin, qmax = quantizer.batch_item(x)
                    watch = (quantizer.__dict__, tuple((k, quantizer.__dict__[k], -1) for k in quantizer._plan_attrs))
                    items.append((xi, y, scale, zp, axis, qmin, qmax, watch))
            else:                                           # LUT: float32 results whatever the input's type
                self.outputs = [torch.empty(x.shape, dtype=torch.float32, device=x.device) for x in self.inputs]
                d = quantizer.__dict__
                watch = (d, (("_stale", d.get("_stale"), -1), ("_lut_table_torch", d.get("_lut_table_torch"), -1)))
                for x, y in zip(self.inputs, self.outputs):
                    item = quantizer.batch_item_lut(x)
                    items.append(item[:2] + (y,) + item[3:] + (watch,))
            self._plan = ops._fast_mod().BatchPlan(items)
            self._plan()
            return
        self.mode = "graph"
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(max(1, warmup)):                 # allocator, lazy library state, launch plans
                for x in self.inputs:
                    fn(x)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with no_gc_while_capturing(), torch.cuda.graph(self.graph), torch.no_grad():
            root = torch.cuda.current_stream()
            branches = [torch.cuda.Stream() for _ in range(self.lanes)] if self.lanes > 1 else [root]
            for b in branches:
                if b is not root:
                    b.wait_stream(root)                     # fork
            for i, x in enumerate(self.inputs):
                with torch.cuda.stream(branches[i % len(branches)]):
                    self.outputs[i] = fn(x)
            for b in branches:
                if b is not root:
                    root.wait_stream(b)                     # join

    def run(self) -> List[torch.Tensor]:
        """Quantize whatever ``inputs`` hold now into ``outputs`` (valid until the next replay)."""
        if self._plan is not None:
            if self._plan() is NotImplemented:              # the quantizer's parameters changed: eager calls from now on
                self._plan = None
                self.mode = "eager"
                return self.run()
        elif self.graph is not None:
            self.graph.replay()
        else:
            with torch.no_grad():
                self.outputs = [self.fn(x) for x in self.inputs]
        return self.outputs

    def fits(self, batches: Sequence[torch.Tensor]) -> bool:
        if len(batches) != self.depth:
            return False
        ref = self.inputs[0]
        return all(isinstance(b, torch.Tensor) and b.shape == ref.shape and b.dtype == ref.dtype
                   and b.device == ref.device and not b.requires_grad for b in batches)

    def __call__(self, batches: Sequence[torch.Tensor]) -> List[torch.Tensor]:
        """``depth`` batches of the captured shape: copy in (skipped for tensors that ARE the static inputs), replay.
        Anything else -- another shape, dtype, device or count, tensors that require grad -- runs the eager calls."""
        if not self.fits(batches):
            return [self.fn(b) for b in batches]
        for dst, src in zip(self.inputs, batches):
            if src is not dst:
                dst.copy_(src)
        return self.run()

    def release(self):
        self.graph, self._plan = None, None
        self.mode = "eager"


def capture_stream(quantizer_or_holder: Callable[[torch.Tensor], torch.Tensor], example: torch.Tensor, depth: int = 16,
                   lanes: int = 1, mode: str = "auto") -> CapturedStream:
    """``depth`` calls of an activation quantizer (or a holder module) on ``example``-shaped batches per replay: one fused
    batched launch for the affine quantizers, one hipGraph otherwise; see ``CapturedStream``."""
    return CapturedStream(quantizer_or_holder, example, depth, lanes, mode=mode)
