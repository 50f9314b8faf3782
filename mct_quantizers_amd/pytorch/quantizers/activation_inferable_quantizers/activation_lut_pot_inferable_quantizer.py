"""Import-path shim: the reference keeps ActivationLutPOTInferableQuantizer here; the implementation is in mct_quantizers_amd.pytorch.quantizers.lut."""
from mct_quantizers_amd.pytorch.quantizers.lut import ActivationLutPOTInferableQuantizer  # noqa: F401
