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

``accelerate(model, capture=True)`` (``MCTQ_AUTO_CAPTURE=1`` for loaded models; default off) additionally lets the model
replay its forward from one hipGraph per input signature, captured at the signature's second occurrence (``AutoCapture``):
outputs are clones, everything that cannot be replayed runs eagerly.

``accelerate(model, example_inputs=(x,))`` captures the whole forward right away -- weight re-quantization, layers,
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


def accelerate(model: nn.Module, example_inputs: Optional[Sequence[torch.Tensor]] = None, capture: bool = False,
               reuse: Optional[str] = None):
    """Install the one-launch-per-forward weight re-quantization on ``model`` (idempotent) and return the model; with
    ``example_inputs`` capture the whole forward into one hipGraph as well and return the replaying callable; with
    ``capture=True`` let the model itself replay its forward from a hipGraph per input signature (``AutoCapture``).

    ``reuse="versioned"`` (opt-in): quantize once, not per forward -- the launch is SKIPPED while every wrapped weight is
    what the last launch read (same device pointer, same in-place version counter, same dtype and sizes) and the
    quantizers' parameters are unchanged; an optimizer step, ``load_state_dict``, a ``.data`` swap or an edited threshold
    relaunches the whole plan at the next forward.  In-place writes through ``weight.data`` move no version counter: call
    ``accelerated(model).invalidate()`` after those.  An inference model then streams no weight through HBM per forward
    at all (ResNet-50: 204 MB, 34 us saved); the per-forward cost is the check (``BatchPlan``: one C call)."""
    if not isinstance(model, nn.Module):
        raise TypeError("accelerate() takes a torch.nn.Module")
    if reuse not in (None, "versioned"):
        raise ValueError('reuse must be None or "versioned"')
    if reuse == "versioned":
        if example_inputs is not None:
            raise ValueError('reuse="versioned" does not combine with example_inputs (a captured forward replays its launch)')
        handle = model.__dict__.get(_KEY)
        # (getattr: a handle un-pickled from a model saved by a build without versioned reuse has no such attribute)
        if handle is not None and not getattr(handle, "versioned", False):
            # re-install in the versioned form; a forward replay the caller had installed (capture=True) comes back with it
            capture = capture or model.__dict__.get("_mctq_auto_capture") is not None
            decelerate(model)
    if capture and example_inputs is None:
        return auto_capture(model, reuse=reuse)
    if example_inputs is not None:
        from mct_quantizers_amd.pytorch.graphs import capture_forward
        decelerate(model)                                  # the captured forward brings its own (non-auto) batcher
        return capture_forward(model, *example_inputs, strict=False)     # other shapes: the eager forward, not an error
    if model.__dict__.get(_KEY) is None and _has_wrapped_weights(model):
        from mct_quantizers_amd.pytorch.batching import BatchedWeightQuantization, batch_weight_quantization
        # a batcher the caller installed by hand (batch_weight_quantization / capture_forward) already does the work
        if not any(isinstance(getattr(h, "__self__", None), BatchedWeightQuantization)
                   for h in model._forward_pre_hooks.values()):
            model.__dict__[_KEY] = batch_weight_quantization(model, reuse_buffers=True, auto=True,
                                                             versioned=reuse == "versioned")
    return model


def decelerate(model: nn.Module) -> nn.Module:
    """Remove what ``accelerate`` installed: every wrapper calls its own quantizer again."""
    cap = model.__dict__.pop("_mctq_auto_capture", None)
    if cap is not None:
        cap.release()
    handle = model.__dict__.pop(_KEY, None)
    if handle is not None:
        handle.remove()
    return model


def accelerate_loaded(obj):
    """What the loaders call on the object ``torch.load`` gave them.  ``MCTQ_AUTO_BATCH=0`` also REMOVES the hook from a
    model that was saved with it (a re-saved accelerated model): the switch means the reference's per-layer calls."""
    if isinstance(obj, nn.Module):
        if auto_batch_enabled():
            accelerate(obj, capture=auto_capture_enabled(), reuse=auto_reuse())
        else:
            decelerate(obj)
    return obj


def auto_reuse() -> Optional[str]:
    """``MCTQ_AUTO_REUSE=versioned`` (default: unset): loaded models get ``accelerate(model, reuse="versioned")`` -- no weight
    launch at all while no weight changed.  Opt-in because of its one blind spot (in-place writes through ``weight.data``)."""
    v = os.environ.get("MCTQ_AUTO_REUSE", "").strip().lower()
    if v in ("", "0", "off", "false", "no"):
        return None
    if v != "versioned":
        raise ValueError('MCTQ_AUTO_REUSE must be unset or "versioned"')
    return "versioned"


# ---------------------------------------------------------------------------------------------------------------------
# the forward replayed from a hipGraph without the caller doing anything: accelerate(model, capture=True) / MCTQ_AUTO_CAPTURE=1
# ---------------------------------------------------------------------------------------------------------------------

def auto_capture_enabled() -> bool:
    """``MCTQ_AUTO_CAPTURE`` (default OFF): loaded models also get ``accelerate(model, capture=True)``."""
    return os.environ.get("MCTQ_AUTO_CAPTURE", "0").strip().lower() not in ("0", "off", "false", "no", "")


class AutoCapture:
    """``model.forward`` replaced (on the instance) by a dispatcher that replays the ORIGINAL forward from one hipGraph per
    input signature.  At batch 1 an eager forward of a wrapped ResNet-50 spends 2.5 ms in torch's dispatch for 1 ms of GPU
    work (profiles/r04/bench_e2e_resnet50.json): the replay removes that.

    What is replayed is the forward BEHIND the module's hooks: ``model(x)`` still runs its forward pre-hooks eagerly -- among
    them the batched re-quantization of all wrapped weights into their persistent buffers (``accelerate``), so weight updates
    are followed exactly as in eager mode -- and then, instead of the Python forward, the graph that reads those buffers.

    A signature (shapes, dtypes, devices of the positional tensor arguments) is captured at its SECOND occurrence; anything
    else runs eagerly: keyword arguments, non-tensor or CPU arguments, training mode, grad mode on while an input or any
    parameter of the model requires grad (a replay builds no autograd graph: use ``torch.no_grad()`` for inference), an
    active trace / compile, a model whose weights are not served by the pre-packed plan.  Outputs are CLONES of
    the graph's static buffers (the caller may keep them).  A rebuilt plan (changed weights-quantizer parameters,
    ``model.half()``), a swapped or re-parameterised activation quantizer, a toggled ``quantization_bypass``
    (``_fingerprint``) or ``release()`` drop the graphs; what a replay cannot follow is a change of the module tree itself
    or of Python state the forward reads that none of these cover.

    Hooks do not fire inside a replay, and during warm-up and capture they would fire three extra times and keep tensors
    of the graph's private pool: while ANY sub-module carries a forward / forward-pre / backward hook, or any GLOBAL module
    hook is registered (``register_module_forward_hook`` ...), every call is eager -- whenever the hook was registered,
    before the first capture or after it.  The signature also carries the autocast state (a graph captured outside
    ``torch.autocast`` is not replayed inside it) and every argument's strides (a channels_last input gets its own graph,
    with channels_last static buffers); all arguments must live on ONE device, which is made current for warm-up, capture
    and replay."""

    def __init__(self, model: nn.Module, max_graphs: int = 8):
        self.model = model
        self.max_graphs = max_graphs
        self._prev = model.__dict__.get("forward")      # an instance-level forward somebody else put there (restored by release)
        self._orig = model.forward                      # the class's forward, bound (or that override)
        self._state = None                              # _fingerprint() at the first capture
        self._graphs = {}                               # signature -> (graph, static inputs, static outputs, plan identity)
        self._seen = {}                                 # signature -> occurrences before capture (or -1: never capture)
        self._busy = False
        model.forward = self._dispatch                  # instance attribute: nn.Module._call_impl calls self.forward

    def __getstate__(self):                             # pickled with the model (torch.save): graphs stay behind
        return {"model": self.model, "max_graphs": self.max_graphs, "_prev": self.__dict__.get("_prev")}

    def __setstate__(self, state):
        self.model, self.max_graphs, self._prev = state["model"], state["max_graphs"], state.get("_prev")
        self._orig = self._prev if self._prev is not None else type(self.model).forward.__get__(self.model)
        self._graphs, self._seen, self._busy, self._state = {}, {}, False, None

    def _modules(self):
        mods = self.__dict__.get("_mods")
        if mods is None:
            from mct_quantizers_amd.pytorch.containers import PytorchActivationQuantizationHolder
            allm = [m for m in self.model.modules() if m is not self.model]
            mods = self.__dict__["_mods"] = (allm, [m for m in allm if isinstance(m, PytorchActivationQuantizationHolder)])
        return mods

    def _hooked(self) -> bool:
        """Is there any hook a replay would skip (or warm-up and capture would fire three extra times)?  Sub-module hooks of
        every kind and the process-wide module hooks; the root's own hooks run eagerly around the replay and do not count."""
        g = torch.nn.modules.module
        if g._global_forward_hooks or g._global_forward_pre_hooks or g._global_backward_hooks or g._global_backward_pre_hooks:
            return True
        for m in self._modules()[0]:
            if m._forward_hooks or m._forward_pre_hooks or m._backward_hooks or m._backward_pre_hooks:
                return True
        return False

    def _fingerprint(self):
        """Python-level state a replay would otherwise freeze: every holder's quantizer object, its launch state (assigning
        to a public parameter drops it) and bypass switch.  Compared on every call, a difference drops the graphs."""
        mods = self._modules()
        state = []
        for h in mods[1]:
            q = h.__dict__.get("activation_holder_quantizer")
            state.append((id(q), id(getattr(q, "__dict__", {}).get("_plan")), bool(h.__dict__.get("quantization_bypass"))))
        return tuple(state)

    def _signature(self, args, kwargs):
        if kwargs or not args or self._busy or self.model.training:
            return None
        if torch.jit.is_tracing() or torch.compiler.is_compiling():
            return None
        grad = torch.is_grad_enabled()
        if grad:                                        # a replay builds no autograd graph: only when nothing could ask for one
            params = self.__dict__.get("_params")
            if params is None:
                params = self.__dict__["_params"] = list(self.model.parameters())
            for p in params:
                if p.requires_grad:
                    return None
        sig = []
        dev = None
        for a in args:
            if type(a) is not torch.Tensor or not a.is_cuda or (grad and a.requires_grad):
                return None
            if dev is None:
                dev = a.device.index
            elif a.device.index != dev:                 # capture and replay bind to ONE current device
                return None
            sig.append((tuple(a.shape), tuple(a.stride()), a.dtype))
        handle = accelerated(self.model)
        if handle is None or handle._plan is None:      # the graph reads the plan's persistent weight buffers
            return None
        if self._hooked():                              # hooks do not fire in a replay: eager while any is registered
            if self._graphs:
                self._graphs.clear()
                self._seen.clear()
            return None
        # autocast changes what the forward computes (dtypes of the matmuls / convolutions): part of the signature
        ac = (torch.is_autocast_enabled("cuda"), torch.get_autocast_dtype("cuda")) if torch.is_autocast_enabled("cuda") else None
        return (dev, ac) + tuple(sig)

    def _dispatch(self, *args, **kwargs):
        sig = self._signature(args, kwargs)
        if sig is None:
            return self._orig(*args, **kwargs)
        plan = accelerated(self.model)._plan
        hit = self._graphs.get(sig)
        if self._graphs:
            if next(iter(self._graphs.values()))[3] is not plan or self._fingerprint() != self._state:
                self._graphs.clear()                    # the weights moved to other buffers / Python-level state changed
                self._seen.clear()
                hit = None
        if hit is None:
            n = self._seen.get(sig, 0)
            if n < 0 or len(self._graphs) >= self.max_graphs:
                return self._orig(*args)
            self._seen[sig] = n + 1
            if n == 0:                                  # first occurrence: eager (warms the allocator and lazy state too)
                return self._orig(*args)
            hit = self._capture(sig, args, plan)
            if hit is None:
                return self._orig(*args)
        graph, static_in, static_out, _ = hit
        try:
            with torch.cuda.device(sig[0]):             # the graph belongs to the arguments' device, current or not
                for dst, src in zip(static_in, args):
                    dst.copy_(src)
                graph.replay()
                return torch.utils._pytree.tree_map(lambda t: t.clone() if isinstance(t, torch.Tensor) else t, static_out)
        except Exception:                               # noqa: BLE001 -- a replay that cannot run: this signature stays eager
            if os.environ.get("MCTQ_CAPTURE_DEBUG"):
                raise
            self._graphs.pop(sig, None)
            self._seen[sig] = -1
            return self._orig(*args)

    def _weights_off_device(self, idx) -> bool:
        from mct_quantizers_amd.pytorch.containers import PytorchQuantizationWrapper
        for m in self._modules()[0]:
            if isinstance(m, PytorchQuantizationWrapper) and m._weights_vars:
                w = m._weights_vars[0][1]
                if isinstance(w, torch.Tensor) and (not w.is_cuda or w.device.index != idx):
                    return True
        return False

    def _capture(self, sig, args, plan):
        self._busy = True
        try:
            if self._weights_off_device(sig[0]):
                raise RuntimeError("the model's weights are not on the arguments' device")
            # streams, graphs and replays bind to the CURRENT device: make it the arguments' one (the model may live on
            # cuda:1 while cuda:0 is current -- its kernels would otherwise go to a stream nobody captures)
            with torch.cuda.device(sig[0]):
                static_in = tuple(a.detach().clone() for a in args)     # clone keeps the strides of a dense argument
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                handle = accelerated(self.model)
                with torch.cuda.stream(side), torch.no_grad():
                    for _ in range(2):
                        handle.reopen()                  # the wrappers take the plan's persistent tensors, as under the hook
                        self._orig(*static_in)
                torch.cuda.current_stream().wait_stream(side)
                torch.cuda.synchronize()
                graph = torch.cuda.CUDAGraph()
                handle.reopen()
                from mct_quantizers_amd.pytorch.graphs import no_gc_while_capturing
                with no_gc_while_capturing(), torch.cuda.graph(graph), torch.no_grad():
                    static_out = self._orig(*static_in)
                if accelerated(self.model)._plan is not plan:
                    raise RuntimeError("the plan changed during capture")
                for t in torch.utils._pytree.tree_leaves(static_out):
                    if isinstance(t, torch.Tensor) and t.is_cuda and t.device.index != sig[0]:
                        raise RuntimeError("an output lives on another device than the arguments")
            hit = (graph, static_in, static_out, plan)
            if not self._graphs:
                self._state = self._fingerprint()       # taken AFTER the capture: the eager calls re-made the launch states
            self._graphs[sig] = hit
            return hit
        except Exception:                               # noqa: BLE001 -- not capturable (data-dependent control flow, ...): eager
            if os.environ.get("MCTQ_CAPTURE_DEBUG"):
                raise
            self._seen[sig] = -1
            torch.cuda.synchronize()
            return None
        finally:
            self._busy = False

    def release(self):
        """Drop the graphs and give the model its own forward back."""
        self._graphs.clear()
        if self.model.__dict__.get("forward") is not None:
            del self.model.__dict__["forward"]
        if self.__dict__.get("_prev") is not None:
            self.model.__dict__["forward"] = self._prev


_CAPTURE_KEY = "_mctq_auto_capture"


def auto_capture(model: nn.Module, reuse: Optional[str] = None) -> nn.Module:
    """``accelerate(model)`` + replay of the forward from one hipGraph per input signature (``AutoCapture``); idempotent."""
    accelerate(model, reuse=reuse)
    if model.__dict__.get(_CAPTURE_KEY) is None and accelerated(model) is not None:
        model.__dict__[_CAPTURE_KEY] = AutoCapture(model)
    return model
