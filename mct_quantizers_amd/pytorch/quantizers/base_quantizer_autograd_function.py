"""Import-path shim: the reference keeps BaseQuantizerAutogradFunction here; the implementation is in mct_quantizers_amd.pytorch.quantizers.onnx_export."""
from mct_quantizers_amd.pytorch.quantizers.onnx_export import BaseQuantizerAutogradFunction  # noqa: F401
