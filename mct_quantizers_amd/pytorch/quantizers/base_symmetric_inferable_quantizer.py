"""Import-path shim: the reference keeps BaseSymmetricInferableQuantizer here; the implementation is in mct_quantizers_amd.pytorch.quantizers.affine."""
from mct_quantizers_amd.pytorch.quantizers.affine import BaseSymmetricInferableQuantizer  # noqa: F401
