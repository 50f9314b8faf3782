from mct_quantizers_amd.common import constants
from mct_quantizers_amd.common.registry import (BaseInferableQuantizer, QuantizationMethod, QuantizationTarget,
                                               QuantizerID, get_all_subclasses, get_inferable_quantizer_class,
                                               mark_quantizer)
