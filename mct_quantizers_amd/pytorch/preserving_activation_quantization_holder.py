"""Import-path shim: the reference keeps PytorchPreservingActivationQuantizationHolder here; the implementation is in mct_quantizers_amd.pytorch.containers."""
from mct_quantizers_amd.pytorch.containers import PytorchPreservingActivationQuantizationHolder  # noqa: F401
