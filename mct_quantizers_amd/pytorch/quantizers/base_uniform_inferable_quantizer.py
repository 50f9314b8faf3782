"""Import-path shim: the reference keeps BaseUniformInferableQuantizer here; the implementation is in mct_quantizers_amd.pytorch.quantizers.affine."""
from mct_quantizers_amd.pytorch.quantizers.affine import BaseUniformInferableQuantizer  # noqa: F401
