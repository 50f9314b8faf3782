"""Loading whole pickled modules (the way MCT ships quantized PyTorch models).

Same entry point as the reference's pytorch/load_model.py:23-34, a pass-through to ``torch.load``.
torch >= 2.6 defaults to ``weights_only=True``, which cannot unpickle modules, so the default here is
``weights_only=False`` unless the caller says otherwise.
"""
import torch


def pytorch_load_quantized_model(filepath, **kwargs):
    kwargs.setdefault("weights_only", False)
    return torch.load(filepath, **kwargs)
