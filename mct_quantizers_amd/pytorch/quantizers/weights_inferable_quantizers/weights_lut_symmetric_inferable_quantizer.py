"""Import-path shim: the reference keeps WeightsLUTSymmetricInferableQuantizer here; the implementation is in mct_quantizers_amd.pytorch.quantizers.lut."""
from mct_quantizers_amd.pytorch.quantizers.lut import WeightsLUTSymmetricInferableQuantizer  # noqa: F401
from mct_quantizers_amd.pytorch.quantizers.onnx_export import WeightsLUTSymmetricF  # noqa: F401,E402  (export branch)
