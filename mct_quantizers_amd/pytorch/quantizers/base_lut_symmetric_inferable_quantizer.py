"""Import-path shim: the reference keeps BaseLUTSymmetricInferableQuantizer here; the implementation is in mct_quantizers_amd.pytorch.quantizers.lut."""
from mct_quantizers_amd.pytorch.quantizers.lut import BaseLUTSymmetricInferableQuantizer  # noqa: F401
