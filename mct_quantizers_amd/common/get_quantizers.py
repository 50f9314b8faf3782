"""Import-path shim: the reference keeps get_inferable_quantizer_class here; the implementation is in mct_quantizers_amd.common.registry."""
from mct_quantizers_amd.common.registry import get_inferable_quantizer_class  # noqa: F401
