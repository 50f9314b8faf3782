"""Import-path shim: the reference keeps ActivationPOTInferableQuantizer here; the implementation is in mct_quantizers_amd.pytorch.quantizers.affine."""
from mct_quantizers_amd.pytorch.quantizers.affine import ActivationPOTInferableQuantizer  # noqa: F401
from mct_quantizers_amd.pytorch.quantizers.onnx_export import ActivationPOTF  # noqa: F401,E402  (export branch)
