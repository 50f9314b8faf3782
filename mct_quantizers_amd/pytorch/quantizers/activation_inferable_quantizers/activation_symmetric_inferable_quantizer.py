"""Import-path shim: the reference keeps ActivationSymmetricInferableQuantizer here; the implementation is in mct_quantizers_amd.pytorch.quantizers.affine."""
from mct_quantizers_amd.pytorch.quantizers.affine import ActivationSymmetricInferableQuantizer  # noqa: F401
