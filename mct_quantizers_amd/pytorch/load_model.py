"""Loading whole pickled modules (the way MCT ships quantized PyTorch models).

Same entry point as the reference's pytorch/load_model.py:23-34, a pass-through to ``torch.load``.
torch >= 2.6 defaults to ``weights_only=True``, which cannot unpickle modules, so the default here is
``weights_only=False`` unless the caller says otherwise.

A loaded module whose wrappers hold weights gets ``accelerate`` applied (``pytorch/accelerate.py``): on a GPU its
forward re-quantizes all wrapped weights in ONE launch instead of one per weight (quantize_wrapper.py:228-240), with
the same results; on the CPU nothing changes.  ``MCTQ_AUTO_BATCH=0`` turns that off.
"""
import torch

from mct_quantizers_amd.pytorch.accelerate import accelerate_loaded


def pytorch_load_quantized_model(filepath, **kwargs):
    kwargs.setdefault("weights_only", False)
    return accelerate_loaded(torch.load(filepath, **kwargs))
