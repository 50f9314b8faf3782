"""Quantizer taxonomy and class registry.

API-compatible with mct_quantizers/common/base_inferable_quantizer.py:21-91,
common/quant_info.py:19-38 and common/get_quantizers.py:22-53: MCT discovers a quantizer by
walking the subclasses of a base class and matching the three attributes stamped by
``@mark_quantizer``; exactly one class must match.
"""
from enum import Enum
from typing import Any, Dict, List, Set

from mct_quantizers_amd.common.constants import QUANTIZATION_METHOD, QUANTIZATION_TARGET, QUANTIZER_ID
from mct_quantizers_amd.logger import Logger


class QuantizationMethod(Enum):
    """Quantization function families (numeric values as in the reference enum)."""
    POWER_OF_TWO = 0
    LUT_POT_QUANTIZER = 1
    SYMMETRIC = 2
    UNIFORM = 3
    LUT_SYM_QUANTIZER = 4


class QuantizationTarget(Enum):
    Activation = "Activation"
    Weights = "Weights"


class QuantizerID(Enum):
    INFERABLE = "inferable_quantizer_id"


def mark_quantizer(quantization_target: QuantizationTarget = None,
                   quantization_method: List[QuantizationMethod] = None,
                   identifier: Any = None):
    """Class decorator: stamp target / supported methods / identifier on a quantizer class."""
    def _stamp(cls):
        setattr(cls, QUANTIZATION_TARGET, quantization_target)
        setattr(cls, QUANTIZATION_METHOD, quantization_method)
        setattr(cls, QUANTIZER_ID, identifier)
        return cls
    return _stamp


class BaseInferableQuantizer:
    """Root of every inferable quantizer (framework independent)."""

    def __init__(self):
        pass

    def initialize_quantization(self, tensor_shape: Any, name: str, layer: Any) -> Dict[Any, Any]:
        """Containers call this when they adopt the quantizer; inferable quantizers own no variables."""
        return {}


def get_all_subclasses(cls: type) -> Set[type]:
    """Transitive closure of ``cls.__subclasses__()``."""
    found: Set[type] = set()
    stack = list(cls.__subclasses__())
    while stack:
        c = stack.pop()
        if c not in found:
            found.add(c)
            stack.extend(c.__subclasses__())
    return found


def get_inferable_quantizer_class(quant_target: QuantizationTarget,
                                  quant_method: QuantizationMethod,
                                  quantizer_base_class: type) -> type:
    """The single inferable quantizer class under ``quantizer_base_class`` for (target, method)."""
    matches = [c for c in get_all_subclasses(quantizer_base_class)
               if getattr(c, QUANTIZATION_TARGET, None) == quant_target
               and getattr(c, QUANTIZATION_METHOD, None) is not None
               and quant_method in getattr(c, QUANTIZATION_METHOD)
               and getattr(c, QUANTIZER_ID, None) is QuantizerID.INFERABLE]
    if len(matches) != 1:
        Logger.error(f"Found {len(matches)} quantizer for target {quant_target.value} "
                     f"that matches the requested quantization method {quant_method.name} "
                     f"but there should be exactly one."
                     f"The possible quantizers that were found are {matches}.")
    return matches[0]
