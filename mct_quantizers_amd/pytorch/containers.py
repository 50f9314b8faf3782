"""nn.Module containers that CALL the quantizers every forward: the drop-in boundary, caller side.

API-compatible with the reference containers (relative to /root/reference/mct_quantizers/pytorch/):
  quantize_wrapper.py:29-270                              PytorchQuantizationWrapper
  activation_quantization_holder.py:23-63                 PytorchActivationQuantizationHolder
  fln_activation_quantization_holder.py:24-56             PytorchFLNActivationQuantizationHolder
  preserving_activation_quantization_holder.py:24-56      PytorchPreservingActivationQuantizationHolder

Attribute / parameter names (``layer``, ``weights_quantizers``, ``positional_weight_<i>``,
``quantized_positional_weight_<i>``, ``activation_holder_quantizer``, ``quantization_bypass``) are kept
because state dicts and pickles of MCT-exported models refer to them.  No kernels here.
"""
import inspect
from typing import Any, Callable, Dict, List, Tuple, Union

import torch
import torch.nn as nn
import torch.nn.modules.module as _nn_module

from mct_quantizers_amd.common.constants import (ACTIVATION_HOLDER_QUANTIZER, LAYER, POSITIONAL_WEIGHT,
                                                 QUANTIZED_POSITIONAL_WEIGHT, TRAINING)
from mct_quantizers_amd.common.registry import BaseInferableQuantizer
from mct_quantizers_amd.logger import Logger


_TRAINING_FLAG_BY_CLASS = {}


def _takes_training_flag(quantizer) -> bool:
    """Does ``quantizer.__call__`` have a parameter literally named ``training``?  The reference reflects on
    the signature in every forward (quantize_wrapper.py:232); the answer is a property of the class, so it is
    looked up once per class here."""
    cls = type(quantizer)
    flag = _TRAINING_FLAG_BY_CLASS.get(cls)
    if flag is None:
        flag = TRAINING in inspect.signature(quantizer.__call__).parameters
        _TRAINING_FLAG_BY_CLASS[cls] = flag
    return flag


_is_compiling = torch.compiler.is_compiling


class PytorchQuantizationWrapper(nn.Module):
    """Wrap a layer (or a function with constant inputs) and fake-quantize its weights on every forward.

    ``weights_quantizers`` maps a weight attribute name (str) -- or, for functional ops with constant
    operands, the operand position (int) -- to its quantizer.  ``weight_values`` maps positions to the
    constant tensors.  ``op_call_args`` / ``op_call_kwargs`` are appended to every call of the wrapped
    op; ``is_inputs_as_list`` passes the tensors as one list (torch.cat style).
    """

    def __init__(self,
                 module: Union[nn.Module, Callable],
                 weights_quantizers: Dict[Union[int, str], BaseInferableQuantizer],
                 weight_values: Dict[int, torch.Tensor] = None,
                 op_call_args: List = None,
                 op_call_kwargs: Dict[str, Any] = None,
                 is_inputs_as_list: bool = False):
        super().__init__()
        if isinstance(module, nn.Module):
            self.add_module(LAYER, module)
        else:
            setattr(self, LAYER, module)          # plain callable (torch.add, torch.cat, ...)

        self.weights_quantizers = weights_quantizers
        self.weight_values = dict() if weight_values is None else weight_values
        for pos, value in self.weight_values.items():
            if not isinstance(value, torch.Tensor):
                Logger.error(f'Positional weight at position {pos} should be a torch.Tensor, '
                             f'but type is {type(value)}.')
        self.op_call_args = [] if op_call_args is None else op_call_args
        self.op_call_kwargs = {} if op_call_kwargs is None else op_call_kwargs
        self.is_inputs_as_list = is_inputs_as_list

        # Either every weight is a named attribute of the layer, or every weight is positional.
        if len(self.weight_values) == 0:
            if not all(isinstance(k, str) for k in self.weights_quantizers):
                Logger.error('"weights_quantizers" keys should be all strings')
            self.is_str_attr = True
        else:
            if not all(isinstance(k, int) for k in self.weight_values):
                Logger.error('All "weight_values" keys should be integers')
            if not all(a == b for a, b in zip(weights_quantizers, weight_values)):
                Logger.error('Mismatch between "weights_quantizers" and "weight_values" keys')
            self.is_str_attr = False

        self._set_weights_vars(True)

    @property
    def is_weights_quantization(self) -> bool:
        return self.num_weights_quantizers > 0

    @property
    def num_weights_quantizers(self) -> int:
        return len(self.weights_quantizers)

    def convert_to_inferable_quantizers(self):
        """Swap trainable quantizers (objects with ``convert2inferable``) for their inferable form."""
        if not self.is_weights_quantization:
            return
        converted = {}
        for name, quantizer in self.weights_quantizers.items():
            conv = getattr(quantizer, 'convert2inferable', None)
            if callable(conv):
                converted[name] = conv()
        self.weights_quantizers = converted
        self._set_weights_vars(False)

    def _set_weights_vars(self, is_training: bool = True):
        """Move the float weights under the wrapper and record (name, weight, quantizer) triples."""
        self._weights_vars = []
        for name, quantizer in self.weights_quantizers.items():
            if self.is_str_attr:
                source = self.layer if is_training else self
                weight = getattr(source, name).detach()
                delattr(self.layer, name)
                setattr(self.layer, name, weight)             # plain tensor now, replaced every forward
                if is_training:
                    self.register_parameter(name, torch.nn.Parameter(weight, requires_grad=True))
                weight_var = getattr(self, name)
            else:
                weight = self.weight_values[name]
                self.register_parameter(f'{POSITIONAL_WEIGHT}_{name}', torch.nn.Parameter(weight, requires_grad=False))
                setattr(self, f'{QUANTIZED_POSITIONAL_WEIGHT}_{name}', weight)
                weight_var = getattr(self, f'{POSITIONAL_WEIGHT}_{name}')
            quantizer.initialize_quantization(weight.shape, name, self)
            self._weights_vars.append((name, weight_var, quantizer))

    def set_quantize_weights(self, quantized_weights: dict):
        """Install freshly quantized weights where the wrapped op will read them.

        ``_set_weights_vars`` left every target as a PLAIN attribute (the layer's parameter was deleted and a tensor
        put in its place), for which ``nn.Module.__setattr__`` ends in ``object.__setattr__`` after ~3 us of
        parameter / buffer / module bookkeeping per weight per forward; writing the instance dictionary is the same
        assignment.  Anything else (the attribute is not a plain one any more) goes through ``setattr``."""
        for key in self.weights_quantizers:
            value = quantized_weights.get(key)
            if self.is_str_attr:
                owner, name = self.layer, key
            else:
                owner, name = self, f'{QUANTIZED_POSITIONAL_WEIGHT}_{key}'
            d = None if _is_compiling() else getattr(owner, "__dict__", None)
            if d is not None and name in d and isinstance(value, torch.Tensor) and not isinstance(value, nn.Parameter):
                d[name] = value
            else:
                setattr(owner, name, value)

    def get_weights_vars(self) -> List[Tuple[str, Any, BaseInferableQuantizer]]:
        return self._weights_vars

    def __getstate__(self):
        # tensors a batched launch prepared for one forward (pytorch/batching.py) are not part of the module's state
        state = super().__getstate__()            # nn.Module's own: leaves out what a .compile()d module cannot pickle
        state.pop("_prequantized_plan", None)
        state.pop("_prequantized_seen", None)
        return state

    def forward(self, *args: List[Any], **kwargs: Dict[str, Any]) -> Union[torch.Tensor, List[torch.Tensor]]:
        if self._weights_vars:
            # tensors a batched launch has already prepared for THIS forward (pytorch/batching.py); used once
            d = self.__dict__
            # (torch.compile traces the plain per-layer calls: none of this bookkeeping belongs in a graph)
            ready = None
            if not _is_compiling():
                # (generation cell [generation, open], {name: tensor}); valid once per generation and only while the
                # model forward whose pre-hook prepared the tensors is still running
                plan = d.get("_prequantized_plan")
                if plan is not None and plan[0][1] and plan[0][0] != d.get("_prequantized_seen"):
                    d["_prequantized_seen"] = plan[0][0]
                    ready = plan[1]
            fresh = {}
            for name, weight, quantizer in self._weights_vars:
                if ready is not None and name in ready:
                    fresh[name] = ready[name]
                elif _takes_training_flag(quantizer):
                    fresh[name] = quantizer(weight, self.training)
                else:
                    fresh[name] = quantizer(weight)
            self.set_quantize_weights(fresh)

        if not self.is_str_attr:
            # constants take their recorded operand positions among the runtime inputs
            args = list(args)
            for pos in sorted(w[0] for w in self._weights_vars):
                args.insert(pos, getattr(self, f'{QUANTIZED_POSITIONAL_WEIGHT}_{pos}'))

        if self.op_call_kwargs or self.op_call_args or kwargs or self.is_inputs_as_list:
            call_kwargs = {**self.op_call_kwargs, **kwargs}
            if self.is_inputs_as_list:
                return self.layer(args, *self.op_call_args, **call_kwargs)
            return self.layer(*args, *self.op_call_args, **call_kwargs)
        return self.layer(*args)

    def get_quantized_weights(self) -> Dict[str, torch.Tensor]:
        return {name: quantizer(w) for name, w, quantizer in self.get_weights_vars()}


def _is_own_call(qtype) -> bool:
    """Is ``qtype.__call__`` the ``__call__`` one of this package's own per-tensor activation quantizers defines?"""
    own = _OWN_CALLS.get("v")
    if own is None:
        from mct_quantizers_amd.pytorch.quantizers import affine
        own = _OWN_CALLS["v"] = {c.__dict__["__call__"] for c in vars(affine).values()
                                 if isinstance(c, type) and c.__module__ == affine.__name__ and "__call__" in c.__dict__}
    call = getattr(qtype, "__call__", None)
    return call in own


_OWN_CALLS = {}


class PytorchActivationQuantizationHolder(torch.nn.Module):
    """Module that owns one activation quantizer and applies it to whatever flows through."""

    def __init__(self, activation_holder_quantizer: BaseInferableQuantizer, **kwargs):
        super().__init__(**kwargs)
        self.activation_holder_quantizer = activation_holder_quantizer
        self.activation_holder_quantizer.initialize_quantization(None, ACTIVATION_HOLDER_QUANTIZER + "_out", self)

    def forward(self, inputs):
        return self.activation_holder_quantizer(inputs)

    def __call__(self, inputs, *args, **kwargs):
        """``nn.Module.__call__`` costs ~2.5 us of Python before ``forward`` runs -- as much as the launch itself for
        a small activation.  When nothing that machinery serves is present (no hooks on this module, no global
        hooks, no compiled call, no torch.jit trace, a plain tensor argument) the result of ``Module.__call__`` IS
        ``forward(inputs)``, so go there directly; anything else takes the full path.

        For the affine activation quantizers even that test, ``forward`` and the quantizer's own ``__call__`` are one C
        call (the compiled binding's ``HolderCall``: the same conditions checked on the dictionaries themselves, then
        the quantizer's pre-packed launch); it answers NotImplemented whenever any of them fails."""
        fast = self.__dict__.get("_fast_call")
        if fast is not None and not args and not kwargs and not _is_compiling():
            y = fast(inputs)
            if y is not NotImplemented:
                return y
        if (type(inputs) is not torch.Tensor or args or kwargs
                or self._forward_hooks or self._forward_pre_hooks or self._backward_hooks or self._backward_pre_hooks
                or _nn_module._global_forward_hooks or _nn_module._global_forward_pre_hooks
                or _nn_module._global_backward_hooks or _nn_module._global_backward_pre_hooks
                or self._compiled_call_impl is not None or torch._C._get_tracing_state() is not None):
            return super().__call__(inputs, *args, **kwargs)
        y = self.forward(inputs)
        if inputs.is_cuda:                                   # (the quantizer's own call above re-made its plan if needed)
            key = self.__dict__.get("_fast_key")
            q = self.__dict__.get(ACTIVATION_HOLDER_QUANTIZER)
            if key is None or key[0] is not q or key[1] is not getattr(q, "__dict__", {}).get("_plan"):
                self._make_fast_call()
        return y

    def _make_fast_call(self):
        """Build the one-call form for the quantizer in place now (False = not available: no compiled binding, a LUT or
        foreign quantizer, a subclass with its own ``forward``).  Re-made whenever the C call reports a change and the
        slow path runs: the quantizer was swapped or one of its parameters assigned."""
        self.__dict__["_fast_call"] = None
        q = self.__dict__.get(ACTIVATION_HOLDER_QUANTIZER)
        plan = getattr(q, "__dict__", {}).get("_plan")
        self.__dict__["_fast_key"] = (q, plan)    # what the C call was (or was not) built for: re-made when it changes
        ok = (type(self).forward in (PytorchActivationQuantizationHolder.forward, _BypassableHolder.forward)
              and type(self).__call__ is PytorchActivationQuantizationHolder.__call__
              and hasattr(q, "_plan_attrs") and plan is not None and plan is not False
              # the C call goes to the plan directly: only when the quantizer's __call__ is one of this package's own (a
              # user subclass that overrides __call__ must be called)
              and _is_own_call(type(q)))
        if ok:
            from mct_quantizers_amd.hip import ops
            fast = ops._fast_mod()
            if fast is not None and type(plan) is getattr(fast, "AffinePlan", None) and hasattr(fast, "HolderCall"):
                hooks = (self._forward_hooks, self._forward_pre_hooks, self._backward_hooks, self._backward_pre_hooks,
                         _nn_module._global_forward_hooks, _nn_module._global_forward_pre_hooks,
                         _nn_module._global_backward_hooks, _nn_module._global_backward_pre_hooks)
                self.__dict__["_fast_call"] = fast.HolderCall(self.__dict__, q, q.__dict__, plan, hooks,
                                                              isinstance(self, _BypassableHolder))

    def __getstate__(self):
        state = super().__getstate__()            # nn.Module's own (drops _compiled_call_impl: a .compile()d holder pickles)
        state.pop("_fast_call", None)             # C object over this module's own dictionaries: rebuilt on first use
        state.pop("_fast_key", None)
        return state

    def capture_stream(self, example: torch.Tensor, depth: int = 16, mode: str = "auto"):
        """Extension (not in the reference): a fixed-shape stream of ``depth`` activation batches through this holder per
        replay -- ONE batched launch for the affine quantizers, one hipGraph otherwise (``pytorch/graphs.py``:
        ``CapturedStream``); falls back to eager calls for anything that does not fit the captured shape."""
        from mct_quantizers_amd.pytorch.graphs import capture_stream
        return capture_stream(self, example, depth, mode=mode)

    def convert_to_inferable_quantizers(self):
        conv = getattr(self.activation_holder_quantizer, 'convert2inferable', None)
        if callable(conv):  # pragma: no cover
            self.activation_holder_quantizer = conv()


class _BypassableHolder(PytorchActivationQuantizationHolder):
    def __init__(self, activation_holder_quantizer: BaseInferableQuantizer, quantization_bypass: bool = False,
                 **kwargs):
        super().__init__(activation_holder_quantizer=activation_holder_quantizer, **kwargs)
        self.quantization_bypass = quantization_bypass

    def forward(self, inputs):
        if self.quantization_bypass:
            return inputs
        return super().forward(inputs)


class PytorchFLNActivationQuantizationHolder(_BypassableHolder):
    """Holder for activations inside fused-layer-norm style blocks; can be switched to pass-through."""


class PytorchPreservingActivationQuantizationHolder(_BypassableHolder):
    """Holder for quantization-preserving ops (reshape, pooling, ...); can be switched to pass-through."""
