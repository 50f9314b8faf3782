"""The nine PyTorch inferable quantizers (same names as mct_quantizers.pytorch.quantizers)."""
from mct_quantizers_amd.pytorch.quantizers.affine import (ActivationPOTInferableQuantizer,
                                                          ActivationSymmetricInferableQuantizer,
                                                          ActivationUniformInferableQuantizer,
                                                          BasePyTorchInferableQuantizer,
                                                          BaseSymmetricInferableQuantizer,
                                                          BaseUniformInferableQuantizer,
                                                          WeightsPOTInferableQuantizer,
                                                          WeightsSymmetricInferableQuantizer,
                                                          WeightsUniformInferableQuantizer)
from mct_quantizers_amd.pytorch.quantizers.lut import (ActivationLutPOTInferableQuantizer,
                                                       BaseLUTSymmetricInferableQuantizer,
                                                       WeightsLUTPOTInferableQuantizer,
                                                       WeightsLUTSymmetricInferableQuantizer)

# the reference's package also exposes its sub-modules as attributes (mct_quantizers.pytorch_quantizers.<module>): the
# import-path shims of the same names are imported here so that such dotted accesses resolve
from mct_quantizers_amd.pytorch.quantizers import (activation_inferable_quantizers,  # noqa: E402,F401
                                                   base_lut_symmetric_inferable_quantizer,
                                                   base_pytorch_inferable_quantizer, base_quantizer_autograd_function,
                                                   base_symmetric_inferable_quantizer, base_uniform_inferable_quantizer,
                                                   weights_inferable_quantizers)
