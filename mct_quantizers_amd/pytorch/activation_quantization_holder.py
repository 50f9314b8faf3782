"""Import-path shim: the reference keeps PytorchActivationQuantizationHolder here; the implementation is in mct_quantizers_amd.pytorch.containers."""
from mct_quantizers_amd.pytorch.containers import PytorchActivationQuantizationHolder  # noqa: F401
