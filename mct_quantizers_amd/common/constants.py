"""Constants shared by the quantizers and containers (values as in mct_quantizers/common/constants.py:82-97)."""

# attribute names stamped on quantizer classes by ``mark_quantizer``
QUANTIZATION_TARGET = 'quantization_target'
QUANTIZATION_METHOD = 'quantization_method'
QUANTIZER_ID = 'identifier'

# container attribute names (state_dict / pickle keys of exported models depend on them)
LAYER = "layer"
TRAINING = "training"
ACTIVATION_HOLDER_QUANTIZER = "activation_holder_quantizer"
POSITIONAL_WEIGHT = 'positional_weight'
QUANTIZED_POSITIONAL_WEIGHT = f'quantized_{POSITIONAL_WEIGHT}'

# LUT quantizer defaults
EPS = 1e-8
LUT_VALUES_BITWIDTH = 8

# constructor keyword names
NUM_BITS = 'num_bits'
SIGNED = 'signed'
THRESHOLD = 'threshold'
PER_CHANNEL = 'per_channel'
MIN_RANGE = 'min_range'
MAX_RANGE = 'max_range'
CHANNEL_AXIS = 'channel_axis'
INPUT_RANK = 'input_rank'
LUT_VALUES = 'lut_values'

FOUND_TORCH = True

# ONNX export (common/constants.py:90,96): custom-op domain and the version attribute of every exported node
ONNX_CUSTOM_OP_DOMAIN = "mct_quantizers"
MCTQ_VERSION = "mctq_version"
REFERENCE_API_VERSION = "1.6.0"          # mct_quantizers.__version__ whose op set the exported nodes target
