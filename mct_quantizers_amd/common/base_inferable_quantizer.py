"""Import-path shim: the reference keeps BaseInferableQuantizer, QuantizationTarget, QuantizerID, mark_quantizer here; the implementation is in mct_quantizers_amd.common.registry."""
from mct_quantizers_amd.common.registry import BaseInferableQuantizer, QuantizationTarget, QuantizerID, mark_quantizer  # noqa: F401
