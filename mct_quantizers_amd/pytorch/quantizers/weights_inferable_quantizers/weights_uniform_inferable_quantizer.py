"""Import-path shim: the reference keeps WeightsUniformInferableQuantizer here; the implementation is in mct_quantizers_amd.pytorch.quantizers.affine."""
from mct_quantizers_amd.pytorch.quantizers.affine import WeightsUniformInferableQuantizer  # noqa: F401
from mct_quantizers_amd.pytorch.quantizers.onnx_export import WeightsUniformF, quantize_uniform_weights_torch  # noqa: F401,E402  (export branch)
