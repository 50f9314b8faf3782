"""Running models that were exported with the REFERENCE package on the MI355X kernels.

Two things keep an already exported MCT model off the HIP path even though the classes here are
API-compatible:

* ``torch.save(model)`` pickles class *paths* such as
  ``mct_quantizers.pytorch.quantize_wrapper.PytorchQuantizationWrapper``.  ``install_reference_aliases()``
  registers those module paths in ``sys.modules`` pointing at this package's modules, so un-pickling yields
  this package's classes; their ``__setstate__`` rebuilds the private launch state the reference objects
  never had (host copies of scalars, LUT decision tables) and moves parameter tensors to the working device.
* ``torch.fx.symbolic_trace(holder)`` of a reference holder inlines the quantizer into a
  ``torch.fake_quantize_per_tensor_affine(x, scale, zero_point, qmin, qmax)`` call_function node with
  constant arguments.  ``route_fx_graph()`` rewrites such nodes to
  ``torch.ops.mctq_amd.fq_per_tensor`` / ``fq_per_channel``, which dispatch per device at run time.
  A traced per-tensor WEIGHTS quantizer records the tensor-qparams overload instead -- scale and zero point are
  1-element tensors lifted to ``_tensor_constant*`` attributes (weights_symmetric_inferable_quantizer.py:147-151)
  -- and becomes ``torch.ops.mctq_amd.fq_per_tensor_tqp``, whose kernel reads them on the device.

``load_reference_model(path)`` does both.
"""
from __future__ import annotations

import importlib.util
import sys
from typing import Dict

import torch
import torch.fx

import mct_quantizers_amd as _pkg
from mct_quantizers_amd.common import constants as _constants
from mct_quantizers_amd import logger as _logger
from mct_quantizers_amd.common import registry as _registry
from mct_quantizers_amd.hip import ops as _ops  # noqa: F401  (registers torch.ops.mctq_amd)
from mct_quantizers_amd.pytorch import containers as _containers
from mct_quantizers_amd.pytorch import load_model as _load_model
from mct_quantizers_amd.pytorch import quantizer_utils as _utils
from mct_quantizers_amd.pytorch import quantizers as _quantizers
from mct_quantizers_amd.pytorch.quantizers import affine as _affine
from mct_quantizers_amd.pytorch.quantizers import lut as _lut

_W = "mct_quantizers.pytorch.quantizers.weights_inferable_quantizers."
_A = "mct_quantizers.pytorch.quantizers.activation_inferable_quantizers."

# reference module path -> module of this package that defines the same public names
REFERENCE_MODULES: Dict[str, object] = {
    "mct_quantizers": _pkg,
    "mct_quantizers.common": _pkg.common,
    "mct_quantizers.common.constants": _constants,
    "mct_quantizers.common.base_inferable_quantizer": _registry,
    "mct_quantizers.common.quant_info": _registry,
    "mct_quantizers.common.get_quantizers": _registry,
    "mct_quantizers.common.get_all_subclasses": _registry,
    "mct_quantizers.logger": _logger,
    "mct_quantizers.pytorch": _pkg.pytorch,
    "mct_quantizers.pytorch.quantizer_utils": _utils,
    "mct_quantizers.pytorch.load_model": _load_model,
    "mct_quantizers.pytorch.quantize_wrapper": _containers,
    "mct_quantizers.pytorch.activation_quantization_holder": _containers,
    "mct_quantizers.pytorch.fln_activation_quantization_holder": _containers,
    "mct_quantizers.pytorch.preserving_activation_quantization_holder": _containers,
    "mct_quantizers.pytorch.quantizers": _quantizers,
    "mct_quantizers.pytorch.quantizers.base_pytorch_inferable_quantizer": _affine,
    "mct_quantizers.pytorch.quantizers.base_symmetric_inferable_quantizer": _affine,
    "mct_quantizers.pytorch.quantizers.base_uniform_inferable_quantizer": _affine,
    "mct_quantizers.pytorch.quantizers.base_lut_symmetric_inferable_quantizer": _lut,
    _W[:-1]: _quantizers,
    _W + "weights_symmetric_inferable_quantizer": _affine,
    _W + "weights_pot_inferable_quantizer": _affine,
    _W + "weights_uniform_inferable_quantizer": _affine,
    _W + "weights_lut_symmetric_inferable_quantizer": _lut,
    _W + "weights_lut_pot_inferable_quantizer": _lut,
    _A[:-1]: _quantizers,
    _A + "activation_symmetric_inferable_quantizer": _affine,
    _A + "activation_pot_inferable_quantizer": _affine,
    _A + "activation_uniform_inferable_quantizer": _affine,
    _A + "activation_lut_pot_inferable_quantizer": _lut,
}


def reference_is_installed() -> bool:
    """True if a real ``mct_quantizers`` distribution is importable (aliases would shadow it)."""
    mod = sys.modules.get("mct_quantizers")
    if mod is not None:
        return mod is not _pkg
    try:
        return importlib.util.find_spec("mct_quantizers") is not None
    except (ImportError, ValueError):
        return False


def install_reference_aliases(force: bool = False) -> None:
    """Make ``import mct_quantizers...`` / un-pickling of reference class paths resolve to this package."""
    if reference_is_installed() and not force:
        raise RuntimeError("the real mct_quantizers package is importable; pass force=True to shadow it")
    for name, module in REFERENCE_MODULES.items():
        sys.modules[name] = module


def remove_reference_aliases() -> None:
    for name, module in REFERENCE_MODULES.items():
        if sys.modules.get(name) is module:
            del sys.modules[name]


def route_fx_graph(gm: torch.fx.GraphModule) -> int:
    """Rewrite inlined ATen fake-quant nodes to the ``mctq_amd`` ops; returns the number of nodes rewritten.

    Python-number qparams -> ``fq_per_tensor``; qparams that are graph values (the tensor-qparams overload: get_attr
    constants of a traced weights quantizer, or any other node) -> ``fq_per_tensor_tqp``; per-channel ->
    ``fq_per_channel``.
    """
    names = ("input", "scale", "zero_point", "quant_min", "quant_max")
    names_pc = ("input", "scale", "zero_point", "axis", "quant_min", "quant_max")
    routed = 0
    for node in list(gm.graph.nodes):
        if node.op != "call_function":
            continue
        if node.target is torch.fake_quantize_per_tensor_affine:
            args = list(node.args) + [node.kwargs[k] for k in names[len(node.args):]]
            if isinstance(args[1], torch.fx.Node) or isinstance(args[2], torch.fx.Node):
                if not (isinstance(args[1], torch.fx.Node) and isinstance(args[2], torch.fx.Node)):
                    continue                              # mixed tensor / number qparams: not an ATen overload
                node.target = torch.ops.mctq_amd.fq_per_tensor_tqp
                node.args = (args[0], args[1], args[2], int(args[3]), int(args[4]))
                node.kwargs = {}
                routed += 1
                continue
            node.target = torch.ops.mctq_amd.fq_per_tensor
            node.args = (args[0], float(args[1]), int(args[2]), int(args[3]), int(args[4]))
            node.kwargs = {}
            routed += 1
        elif node.target is torch.fake_quantize_per_channel_affine:
            args = list(node.args) + [node.kwargs[k] for k in names_pc[len(node.args):]]
            node.target = torch.ops.mctq_amd.fq_per_channel
            node.args = tuple(args)
            node.kwargs = {}
            routed += 1
    if routed:
        gm.graph.lint()
        gm.recompile()
    return routed


def adopt(model: torch.nn.Module) -> torch.nn.Module:
    """Post-load fix-ups on a module tree: route traced sub-graphs, refresh quantizer launch state."""
    for module in model.modules():
        if isinstance(module, torch.fx.GraphModule):
            route_fx_graph(module)
        for value in list(vars(module).values()):
            if isinstance(value, _affine.BasePyTorchInferableQuantizer):
                value._rebuild_launch_state()
        quantizers = getattr(module, "weights_quantizers", None)
        if isinstance(quantizers, dict):
            for q in quantizers.values():
                if isinstance(q, _affine.BasePyTorchInferableQuantizer):
                    q._rebuild_launch_state()
    return model


def load_reference_model(filepath, **kwargs) -> torch.nn.Module:
    """``torch.load`` a module pickled with the reference package and run it on this package's kernels."""
    installed_here = not reference_is_installed()
    if installed_here:
        install_reference_aliases()
    try:
        model = _load_model.pytorch_load_quantized_model(filepath, **kwargs)
    finally:
        if installed_here:
            remove_reference_aliases()
    return adopt(model)
