"""Import-path shim: the reference keeps WeightsLUTPOTInferableQuantizer here; the implementation is in mct_quantizers_amd.pytorch.quantizers.lut."""
from mct_quantizers_amd.pytorch.quantizers.lut import WeightsLUTPOTInferableQuantizer  # noqa: F401
