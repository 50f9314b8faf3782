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
wrappers and quantizer parameters found then; in-place updates of the WEIGHTS are followed (that is the point), and so
are replaced or edited quantizer parameters, ``model.half()`` / ``.to(device)`` and re-pointed storages: the plan
re-checks its tensors and the quantizers' public attributes on every call and is rebuilt when something changed.
Only adding / removing wrappers needs ``handle.refresh()``.  Every launch bumps the in-place version counter of the
persistent outputs it rewrites, so autograd notices a quantized weight saved for a backward and overwritten by a later
forward exactly as it notices any other in-place write.  A forward that autograd may record -- grad mode on and an input
of the model or one of its parameters requiring a gradient -- does NOT get the persistent tensors at all: a wrapped layer
saves its quantized weight for the input gradient, two forwards before one backward (``(model(a) + model(b)).sum()
.backward()``) would then find the first forward's saved weight overwritten by the second, where the reference, which
returns fresh tensors, works.  Such forwards take the fresh-tensor launch below (still one launch per storage type);
forwards under ``torch.no_grad()`` / ``torch.inference_mode()`` or of a fully frozen model keep the one-C-call path.
``versioned=True`` (``accelerate(model, reuse="versioned")``) adds plan-level versioned reuse: a forward whose weights are
what the last launch read issues NO launch at all (the reference's ``enable_reuse_quantizer`` idea,
base_pytorch_inferable_quantizer.py:34-49, without its staleness: an optimizer step, ``load_state_dict``, a ``.data``
swap or an edited quantizer parameter relaunches the whole plan).  The launch is ONE grid per storage type whatever the
number of weights (mctq_fq_batch_pack / mctq_fq_batch_run: the descriptors live in a small device table).
A wrapper uses the tensor prepared for it only inside the model forward whose pre-hook has just filled it (generation
counter, closed again by a forward hook): calling a sub-module directly, past the hook, falls back to that wrapper's
own quantizer call -- in both modes.  54 ResNet-50 weights: 70 us -> host ~4 us + ~35 us
of GPU time (profiles/r02/requant_model_weights_batched.log).

The affine weights quantizers (symmetric / power-of-two / uniform, per tensor or per channel) take part in both modes;
with ``reuse_buffers=True`` the LUT weights quantizers whose codebook has a decision table (every legal default
configuration) join through a second table launch (mctq_lutt_batch_run).  Quantizers with the reuse cache enabled,
trainable quantizers and wrappers of positional (functional) weights keep calling their quantizer as before.
"""
from __future__ import annotations

from typing import List, Tuple

import torch
import torch.nn as nn

from mct_quantizers_amd.hip import ops
from mct_quantizers_amd.pytorch.containers import PytorchQuantizationWrapper


class BatchedWeightQuantization:
    """Handle returned by ``batch_weight_quantization``; ``remove()`` restores per-layer quantization."""

    def __init__(self, model: nn.Module, reuse_buffers: bool = False, auto: bool = False, versioned: bool = False):
        self.model = model
        self.reuse_buffers = reuse_buffers or versioned
        # versioned reuse ("quantize once, not per forward"): the pre-packed plan skips its launch while every
        # planned weight is what the last launch read (device pointer, in-place version counter, dtype, sizes), its outputs
        # were not written by anybody else and the quantizers' parameters are unchanged; any change relaunches the whole
        # plan.  Writes through ``weight.data`` do not move a version counter: ``invalidate()`` after those.
        self.versioned = versioned
        # auto: installed by ``accelerate`` / ``pytorch_load_quantized_model`` on a model nobody asked to batch -- it must
        # never change what a forward does except for the number of launches: weights that are not on a GPU, an
        # active trace or compile, or anything the pre-packed launch cannot take leave the per-layer calls in charge
        self.auto = auto
        self._plan = None                 # (BatchPlan, tensor count) in reuse_buffers mode
        # [generation, open]: a wrapper takes its prepared tensor once per generation and only while the model's
        # forward that prepared it is running (a direct call of a sub-module later must not serve it)
        self._cell = [0, False]
        self._hook = model.register_forward_pre_hook(self._before_forward)
        self._post = model.register_forward_hook(self._after_forward, always_call=True)

    def __getstate__(self):
        # torch.save(model) / copy.deepcopy(model) reach the handle through the hooks: the pre-packed launch (device
        # table, raw pointers) stays behind and is rebuilt at the copy's first forward
        state = dict(self.__dict__)
        state["_plan"] = None
        state["_wrappers"] = None
        state["_params"] = None
        state["_installed"] = None
        state["_cell"] = [0, False]
        state.pop("_auto_failed", None)
        return state

    def _entries(self, with_lut: bool = False) -> List[Tuple[PytorchQuantizationWrapper, str, torch.Tensor, object]]:
        out = []
        wrappers = self.__dict__.get("_wrappers")
        if wrappers is None:              # the walk over the module tree, once (refresh() after adding / removing wrappers)
            wrappers = self.__dict__["_wrappers"] = [
                m for m in self.model.modules()
                if isinstance(m, PytorchQuantizationWrapper) and m.is_weights_quantization and m.is_str_attr]
        for m in wrappers:
            for name, weight, quantizer in m.get_weights_vars():
                if (with_lut and hasattr(quantizer, "batch_item_lut") and not quantizer.enable_reuse
                        and not quantizer.__dict__.get("_versioned_reuse")
                        and not (quantizer._use_custom_impl and torch.jit.is_tracing())
                        and isinstance(weight, torch.Tensor) and quantizer.batch_item_lut(weight.detach()) is not None):
                    out.append((m, name, weight, quantizer))      # LUT weights: pre-packed plan only
                    continue
                if (hasattr(quantizer, "batch_item") and not quantizer.enable_reuse
                        and int(getattr(quantizer, "num_bits", 0)) <= 24
                        and not quantizer.__dict__.get("_versioned_reuse")
                        and not (quantizer._use_custom_impl and torch.jit.is_tracing())):
                    out.append((m, name, weight, quantizer))
        return out

    def _auto_applies(self) -> bool:
        """Auto mode, no plan yet: is this a forward the one-launch path may take at all?  (Cheap: runs on every forward
        of a model that stays on the CPU.)"""
        if torch.jit.is_tracing() or torch.compiler.is_compiling():
            return False
        wrappers = self.__dict__.get("_wrappers")
        if wrappers is None:
            self._entries()
            wrappers = self.__dict__["_wrappers"]
        if not wrappers:
            return False
        dev = None
        for m in wrappers:
            vars_ = m._weights_vars
            if not vars_:
                continue
            d = self._gpu_of(vars_[0][1])
            if d is None:
                return False
            if dev is None:
                dev = d
            elif d != dev:                            # a model spread over several GPUs: one launch cannot serve it
                return False
        return True

    @staticmethod
    def _gpu_of(weight):
        """The GPU a weight lives on, or None (CPU tensor, not a tensor)."""
        return weight.device if isinstance(weight, torch.Tensor) and weight.is_cuda else None

    @staticmethod
    def _watch(quantizer):
        """What the plan checks on every call (in C): the quantizer's public parameters are still the objects, at the
        in-place versions, its launch state was derived from (the reference reads them on every call)."""
        d = quantizer.__dict__
        names = ("scales", "zero_points", "per_channel", "channel_axis", "min_quantized_domain", "max_quantized_domain")
        if not all(k in d for k in names):
            return None
        return (d, tuple((k, d[k], d[k]._version if isinstance(d[k], torch.Tensor) else -1) for k in names))

    def _build_plan(self, entries):
        fast = ops._fast_mod()
        if fast is None or not all(w.is_cuda for _, _, w, _ in entries):
            return None
        items, per_wrapper = [], {}
        for wrapper, name, weight, quantizer in entries:
            weight.requires_grad = False            # the side effect of the reference's weights quantizers
            if not hasattr(quantizer, "batch_item"):                 # a LUT weights quantizer with a decision table
                item = quantizer.batch_item_lut(weight)
                if item is None:
                    return None
                y = torch.empty(item[1].shape, dtype=torch.float32, device=item[1].device)
                d = quantizer.__dict__
                watch = (d, (("_stale", d.get("_stale"), -1), ("_lut_table_torch", d.get("_lut_table_torch"), -1),
                             ("_threshold_torch", d.get("_threshold_torch"), d["_threshold_torch"]._version)))
                items.append(item[:2] + (y,) + item[3:] + (watch,))
                per_wrapper.setdefault(wrapper, {})[name] = y
                continue
            x, scales, zps, axis, qmin, qmax = quantizer.batch_item(weight)
            if quantizer.__dict__.get("_zp_out_of_range"):
                return None                         # the per-layer call raises ATen's message
            if zps is None and axis is None and weight.dtype == torch.float64:
                zps = torch.zeros(1, dtype=torch.int32, device=weight.device)
            if not ops._is_dense(x.detach()):
                return None
            y = torch.empty_like(x.detach())
            items.append((x, y, scales, zps, axis, qmin, qmax, self._watch(quantizer)))
            per_wrapper.setdefault(wrapper, {})[name] = y
        try:
            plan = fast.BatchPlan(items, bool(self.__dict__.get("versioned")))
        except TypeError:
            return None                   # something the pre-packed launch cannot take: per-forward batching instead
        for wrapper, outs in per_wrapper.items():
            wrapper.__dict__["_prequantized_plan"] = (self._cell, outs)
            wrapper.__dict__.pop("_prequantized_seen", None)
        self.__dict__["_installed"] = "plan"
        return plan, len(entries), per_wrapper

    def refresh(self):
        """Rebuild the pre-packed plan at the next forward.  Not needed after changing quantizer parameters or moving /
        casting the model (the plan notices and is rebuilt); needed after adding or removing wrappers."""
        self._drop_plan()

    def invalidate(self):
        """Versioned reuse: the next forward launches whatever the version counters say (call after writing weights
        through ``.data`` or raw pointers, which no counter sees)."""
        plan = self._plan
        if plan is not None:
            plan[0].invalidate()

    def stats(self):
        """(launches, skipped calls) of the current pre-packed plan; (0, 0) without one."""
        plan = self._plan
        return tuple(plan[0].stats()) if plan is not None else (0, 0)

    def _drop_plan(self):
        self._plan = None
        self.__dict__["_wrappers"] = None
        self.__dict__["_params"] = None
        self.__dict__["_installed"] = None
        for m in self.model.modules():
            if isinstance(m, PytorchQuantizationWrapper):
                m.__dict__.pop("_prequantized_plan", None)
                m.__dict__.pop("_prequantized_seen", None)

    def _recorded_by_autograd(self, args) -> bool:
        """May this forward's quantized weights be SAVED for a backward?  Grad mode on and a tensor among the model's
        inputs, or a parameter of the model, requires a gradient (a layer saves its -- constant -- weight whenever its input
        requires one).  The persistent buffers of the pre-packed plan must not be handed to such a forward."""
        if not torch.is_grad_enabled():
            return False
        for a in args or ():
            if isinstance(a, torch.Tensor):
                if a.requires_grad:
                    return True
            elif isinstance(a, (list, tuple)):
                if any(isinstance(t, torch.Tensor) and t.requires_grad for t in a):
                    return True
        params = self.__dict__.get("_params")
        if params is None:                # refresh() / a rebuilt plan re-reads the list; requires_grad itself is read per call
            params = self.__dict__["_params"] = list(self.model.parameters())
        for p in params:
            if p.requires_grad:
                return True
        return False

    def _install(self, per_wrapper, kind):
        """Point every wrapper at the tensors of THIS forward (``kind``: "plan" persistent buffers, "fresh" new tensors); a
        wrapper without an entry loses whatever an earlier forward of the other kind left it."""
        cell = self._cell
        for m in self.__dict__.get("_wrappers") or ():
            ready = per_wrapper.get(m)
            if ready is None:
                m.__dict__.pop("_prequantized_plan", None)
            else:
                m.__dict__["_prequantized_plan"] = (cell, ready)
        self.__dict__["_installed"] = kind

    def quantize_now(self, args=None) -> int:
        """Quantize every participating weight in one batched launch and hand the results to the wrappers.
        Returns the number of tensors quantized.  ``args``: the model's inputs (the forward pre-hook passes them)."""
        cell = self._cell
        if self.auto and self._plan is None and not self._auto_applies():
            return 0
        # (a direct call -- no ``args`` -- is not a forward: nothing it prepares can be saved for a backward)
        if self.reuse_buffers and not torch.jit.is_tracing() and not (args is not None and self._recorded_by_autograd(args)):
            for _ in range(2):                                       # a stale plan is rebuilt once, then given up
                plan = self._plan
                if plan is None:
                    entries = self._entries(with_lut=True)
                    plan = self._plan = self._build_plan(entries) if entries else None
                if plan is None:
                    break
                if plan[0]() is None:                                # ONE C call: tensors re-checked, one launch
                    if self.__dict__.get("_installed") != "plan":    # a recorded forward in between handed out fresh tensors
                        self._install(plan[2], "plan")
                    cell[0] += 1
                    cell[1] = True
                    return plan[1]
                self._drop_plan()                                    # NotImplemented: shapes / dtypes / parameters changed
                if self.auto and not self._auto_applies():           # e.g. the model went back to the CPU
                    return 0
        entries = self._entries()
        if not entries:
            return 0
        items = []
        for _, _, weight, quantizer in entries:
            weight.requires_grad = False            # the side effect of the reference's weights quantizers
            items.append(quantizer.batch_item(weight))
        outs = ops.fq_batched(items)
        per_wrapper = {}
        for (wrapper, name, _, _), y in zip(entries, outs):
            per_wrapper.setdefault(wrapper, {})[name] = y
        self._install(per_wrapper, "fresh")
        cell[0] += 1
        cell[1] = True
        return len(entries)

    def reopen(self) -> bool:
        """The tensors of the last launch count as fresh once more (no launch): for a caller that runs the model's forward
        again behind the hook that filled them -- ``accelerate.AutoCapture`` warming up and capturing.  False without a plan."""
        if self._plan is None:
            return False
        self._cell[0] += 1
        self._cell[1] = True
        return True

    def _before_forward(self, module, args):
        if not self.auto:
            self.quantize_now(args)
            return None
        # auto mode was installed on a model nobody asked to batch: whatever goes wrong here must not become the forward's
        # problem -- the per-layer calls are always there (and raise what the reference would raise, if anything)
        if self.__dict__.get("_auto_failed"):
            return None
        try:
            self.quantize_now(args)
        except Exception as e:  # noqa: BLE001
            self.__dict__["_auto_failed"] = True
            self._cell[1] = False
            try:
                self._drop_plan()
            except Exception:  # noqa: BLE001
                pass
            from mct_quantizers_amd.logger import Logger
            Logger.warning(f"mct_quantizers_amd: the one-launch weight re-quantization stood down for this model "
                           f"({type(e).__name__}: {e}); every wrapper calls its own quantizer (MCTQ_AUTO_BATCH=0 silences this)")
        return None

    def _after_forward(self, module, args, output):
        self._cell[1] = False             # whatever a wrapper did not pick up in this forward is void
        return None

    def remove(self):
        self._hook.remove()
        self._post.remove()
        self._drop_plan()


def batch_weight_quantization(model: nn.Module, reuse_buffers: bool = False, auto: bool = False,
                              versioned: bool = False) -> BatchedWeightQuantization:
    """Install the batched weight re-quantization on ``model`` (a forward pre-hook on the given module).
    ``reuse_buffers``: see the module docstring (persistent output tensors + pre-packed launch).
    ``versioned`` (implies ``reuse_buffers``): skip the launch of a forward whose weights are what the last launch read
    (``BatchedWeightQuantization.versioned``).
    ``auto``: stand aside (per-layer calls, as without the hook) whenever the weights are not on a GPU or a trace /
    compile is running -- the mode ``accelerate`` and ``pytorch_load_quantized_model`` install."""
    return BatchedWeightQuantization(model, reuse_buffers, auto, versioned)
