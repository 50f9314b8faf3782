"""Import-path shim: the reference keeps WeightsPOTInferableQuantizer here; the implementation is in mct_quantizers_amd.pytorch.quantizers.affine."""
from mct_quantizers_amd.pytorch.quantizers.affine import WeightsPOTInferableQuantizer  # noqa: F401
from mct_quantizers_amd.pytorch.quantizers.onnx_export import WeightsPOTF  # noqa: F401,E402  (export branch)
