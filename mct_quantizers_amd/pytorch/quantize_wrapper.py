"""Import-path shim: the reference keeps PytorchQuantizationWrapper here; the implementation is in mct_quantizers_amd.pytorch.containers."""
from mct_quantizers_amd.pytorch.containers import PytorchQuantizationWrapper  # noqa: F401
