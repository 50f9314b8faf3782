"""Import-path shim: the reference keeps WeightsSymmetricInferableQuantizer here; the implementation is in mct_quantizers_amd.pytorch.quantizers.affine."""
from mct_quantizers_amd.pytorch.quantizers.affine import WeightsSymmetricInferableQuantizer  # noqa: F401
from mct_quantizers_amd.pytorch.quantizers.onnx_export import WeightsSymmetricF, quantize_sym_weights_torch  # noqa: F401,E402  (export branch)
