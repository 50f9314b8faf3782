"""Import-path shim: the reference keeps ActivationUniformInferableQuantizer here; the implementation is in mct_quantizers_amd.pytorch.quantizers.affine."""
from mct_quantizers_amd.pytorch.quantizers.affine import ActivationUniformInferableQuantizer  # noqa: F401
from mct_quantizers_amd.pytorch.quantizers.onnx_export import ActivationUniformF, quantize_uniform_activations_torch  # noqa: F401,E402  (export branch)
