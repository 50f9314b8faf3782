#!/usr/bin/env python3
"""Drop-in check (build container only): run the REFERENCE's own PyTorch test-suite (/root/reference/tests/pytorch_tests,
unmodified, read in place) with `mct_quantizers` resolving to THIS package (compat.install_reference_aliases).
Nothing of the reference is copied: its test files are collected from where they lie.  The onnx_export_tests need
onnxruntime-extensions / the metadata module (out of scope) and are skipped.  test_pytorch_load_model.py
imports `onnx` and `mct_quantizers.pytorch.metadata` at module level for ONE of its cases (test_save_and_load_metadata,
ONNX metadata_props plumbing): empty stand-in modules let the file import, that case is deselected by name, and the
torch.save -> pytorch_load_quantized_model cases (:78-221, every quantizer inside a wrapper / holder) run.
On a GPU box the same suite exercises the HIP kernels (the working device is 'cuda')."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
REF_TESTS = "/root/reference/tests"
assert os.path.isdir(REF_TESTS), "the reference checkout is only present in the build container"

import random  # noqa: E402

import numpy as np  # noqa: E402
import pytest  # noqa: E402
import torch  # noqa: E402

# The reference's tests draw UNSEEDED random tensors and assert properties that hold only with high probability (e.g. that
# every codebook entry occurs in every channel slice, test_weights_lut_inferable_quantizer.py:75) -- they fail now and then
# against the reference itself.  Fixed seeds make this run reproducible; the seed is not tuned, 0 is the first one tried.
random.seed(0)
np.random.seed(0)
torch.manual_seed(0)
from mct_quantizers_amd import compat  # noqa: E402

compat.install_reference_aliases(force=False)
import mct_quantizers  # noqa: E402,F401
assert mct_quantizers.__name__ == "mct_quantizers_amd", "the real package is importable: refusing to shadow it"
sys.path.insert(0, os.path.dirname(REF_TESTS))          # so that `tests.pytorch_tests...` imports resolve

import importlib.util  # noqa: E402
import types  # noqa: E402


def _out_of_scope(*_a, **_k):
    raise NotImplementedError("ONNX / metadata plumbing is out of this package's scope (SURVEY.md section 2, rows 8-9)")


if importlib.util.find_spec("onnx") is None:             # not installed here; only the deselected metadata case uses it
    sys.modules["onnx"] = types.ModuleType("onnx")
meta = types.ModuleType("mct_quantizers.pytorch.metadata")
for name in ("add_metadata", "add_onnx_metadata", "get_metadata", "get_onnx_metadata"):
    setattr(meta, name, _out_of_scope)
sys.modules.setdefault("mct_quantizers.pytorch.metadata", meta)
sys.modules.setdefault("mct_quantizers_amd.pytorch.metadata", meta)
args = [os.path.join(REF_TESTS, "pytorch_tests"), "-q", "-p", "no:cacheprovider", "--rootdir", "/tmp",
        "--ignore", os.path.join(REF_TESTS, "pytorch_tests", "onnx_export_tests"),
        "-k", "not test_save_and_load_metadata"] + sys.argv[1:]
sys.exit(pytest.main(args))
