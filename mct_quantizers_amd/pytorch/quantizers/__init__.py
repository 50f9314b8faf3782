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
