"""Import-path shim: the reference keeps ActivationSymmetricInferableQuantizer here; the implementation is in mct_quantizers_amd.pytorch.quantizers.affine."""
from mct_quantizers_amd.pytorch.quantizers.affine import ActivationSymmetricInferableQuantizer  # noqa: F401
from mct_quantizers_amd.pytorch.quantizers.onnx_export import ActivationSymF, quantize_sym_activations_torch  # noqa: F401,E402  (export branch)
