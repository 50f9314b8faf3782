"""Import-path shim: the reference keeps ActivationUniformInferableQuantizer here; the implementation is in mct_quantizers_amd.pytorch.quantizers.affine."""
from mct_quantizers_amd.pytorch.quantizers.affine import ActivationUniformInferableQuantizer  # noqa: F401
