"""Import-path shim: the reference keeps PytorchFLNActivationQuantizationHolder here; the implementation is in mct_quantizers_amd.pytorch.containers."""
from mct_quantizers_amd.pytorch.containers import PytorchFLNActivationQuantizationHolder  # noqa: F401
