"""Import-path shim: the reference keeps get_all_subclasses here; the implementation is in mct_quantizers_amd.common.registry."""
from mct_quantizers_amd.common.registry import get_all_subclasses  # noqa: F401
