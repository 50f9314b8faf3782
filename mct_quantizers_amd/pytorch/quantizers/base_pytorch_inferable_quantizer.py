"""Import-path shim: the reference keeps BasePyTorchInferableQuantizer here; the implementation is in mct_quantizers_amd.pytorch.quantizers.affine."""
from mct_quantizers_amd.pytorch.quantizers.affine import BasePyTorchInferableQuantizer  # noqa: F401
