"""Import-path shim: the reference keeps WeightsLUTPOTInferableQuantizer here; the implementation is in mct_quantizers_amd.pytorch.quantizers.lut."""
from mct_quantizers_amd.pytorch.quantizers.lut import WeightsLUTPOTInferableQuantizer  # noqa: F401
from mct_quantizers_amd.pytorch.quantizers.onnx_export import WeightsLUTPOTF  # noqa: F401,E402  (export branch)
