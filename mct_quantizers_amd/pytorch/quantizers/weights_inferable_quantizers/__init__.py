"""Import-path compatibility with the reference package layout (re-exports only)."""
from mct_quantizers_amd.pytorch.quantizers.weights_inferable_quantizers import (  # noqa: F401  (attributes of the package, as in the reference)
    base_weight_quantizer_autograd_function, weights_lut_pot_inferable_quantizer, weights_lut_symmetric_inferable_quantizer,
    weights_pot_inferable_quantizer, weights_symmetric_inferable_quantizer, weights_uniform_inferable_quantizer)
