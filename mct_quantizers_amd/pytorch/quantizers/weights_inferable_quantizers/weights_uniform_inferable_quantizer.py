"""Import-path shim: the reference keeps WeightsUniformInferableQuantizer here; the implementation is in mct_quantizers_amd.pytorch.quantizers.affine."""
from mct_quantizers_amd.pytorch.quantizers.affine import WeightsUniformInferableQuantizer  # noqa: F401
