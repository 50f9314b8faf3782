"""Import-path shim: the reference keeps QuantizationMethod here; the implementation is in mct_quantizers_amd.common.registry."""
from mct_quantizers_amd.common.registry import QuantizationMethod  # noqa: F401
