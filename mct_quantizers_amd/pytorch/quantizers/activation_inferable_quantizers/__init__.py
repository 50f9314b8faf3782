"""Import-path compatibility with the reference package layout (re-exports only)."""
from mct_quantizers_amd.pytorch.quantizers.activation_inferable_quantizers import (  # noqa: F401  (attributes of the package, as in the reference)
    activation_lut_pot_inferable_quantizer, activation_pot_inferable_quantizer, activation_symmetric_inferable_quantizer,
    activation_uniform_inferable_quantizer, base_activation_quantizer_autograd_function)
