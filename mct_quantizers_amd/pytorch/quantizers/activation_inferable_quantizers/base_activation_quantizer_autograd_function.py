"""Import-path shim (reference layout); implementation: mct_quantizers_amd.pytorch.quantizers.onnx_export."""
from mct_quantizers_amd.pytorch.quantizers.onnx_export import BaseActivationQuantizerAutogradFunction  # noqa: F401
