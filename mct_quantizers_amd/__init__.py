"""mct_quantizers_amd -- MI355X (gfx950) native inferable quantizers with the mct_quantizers API.

Scope: the PyTorch inferable-quantizer hot path of sony/mct_quantizers; the public
names below are the subset of mct_quantizers/__init__.py:16-34 that belongs to that path.
"""
__version__ = "0.1.0"

from mct_quantizers_amd.common import constants
from mct_quantizers_amd.common.registry import (BaseInferableQuantizer, QuantizationMethod, QuantizationTarget,
                                               QuantizerID, get_inferable_quantizer_class, mark_quantizer)
from mct_quantizers_amd.pytorch import quantizers as pytorch_quantizers
from mct_quantizers_amd.pytorch.containers import (PytorchActivationQuantizationHolder,
                                                   PytorchFLNActivationQuantizationHolder,
                                                   PytorchPreservingActivationQuantizationHolder,
                                                   PytorchQuantizationWrapper)
from mct_quantizers_amd.pytorch.load_model import pytorch_load_quantized_model


def __getattr__(name):
    # lazily loaded helpers: mctq.compat (reference pickles / fx routing), mctq.sharded (dim-0 shards + all-gather),
    # mctq.consumers (integer GEMM on the codes)
    if name in ("compat", "sharded", "workloads", "consumers"):
        import importlib
        return importlib.import_module(f"mct_quantizers_amd.{name}")
    if name == "capture_forward":                    # the whole forward replayed from one hipGraph
        from mct_quantizers_amd.pytorch.graphs import capture_forward
        return capture_forward
    if name == "capture_stream":                     # a fixed-shape activation stream replayed from one hipGraph
        from mct_quantizers_amd.pytorch.graphs import capture_stream
        return capture_stream
    if name in ("accelerate", "decelerate", "accelerated"):   # one launch per forward for all wrapped weights, stand-aside form
        from mct_quantizers_amd.pytorch import accelerate as _acc
        return getattr(_acc, name)
    if name == "batch_weight_quantization":          # all wrapped weights of a model in ONE launch per forward
        from mct_quantizers_amd.pytorch.batching import batch_weight_quantization
        return batch_weight_quantization
    raise AttributeError(f"module 'mct_quantizers_amd' has no attribute {name!r}")
