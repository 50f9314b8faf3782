"""Import-path shim: the reference keeps WeightsSymmetricInferableQuantizer here; the implementation is in mct_quantizers_amd.pytorch.quantizers.affine."""
from mct_quantizers_amd.pytorch.quantizers.affine import WeightsSymmetricInferableQuantizer  # noqa: F401
