"""Drop-in check in the build container: the REFERENCE's own PyTorch test-suite, collected unmodified from
/root/reference/tests/pytorch_tests, passes with `mct_quantizers` resolving to this package
(tools/run_reference_tests.py).  Skipped wherever the reference checkout is absent (e.g. the GPU box)."""
import os
import subprocess
import sys

import pytest

from conftest import REPO

REF = "/root/reference/tests/pytorch_tests"


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference checkout exists only in the build container")
@pytest.mark.parametrize("auto_batch", ["1", "0"])
def test_reference_pytorch_suite_passes_against_this_package(tmp_path, auto_batch):
    """With the loader's one-launch-per-forward hook on (the default) and off (MCTQ_AUTO_BATCH=0)."""
    env = dict(os.environ)
    env["MCTQ_AUTO_BATCH"] = auto_batch
    env["PYTHONPATH"] = ""                                   # the real mct_quantizers must NOT be importable
    env["TORCH_FORCE_NO_WEIGHTS_ONLY_LOAD"] = "1"           # its save/load tests predate torch's weights_only default
    r = subprocess.run([sys.executable, os.path.join(REPO, "tools", "run_reference_tests.py")],
                       cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=900)
    tail = r.stdout[-1500:]
    assert r.returncode == 0, tail + r.stderr[-1500:]
    assert " passed" in tail and "failed" not in tail.splitlines()[-1], tail
    # 64 cases + the 9 torch.save -> pytorch_load_quantized_model cases of test_pytorch_load_model.py:78-221 (its one
    # ONNX-metadata case is deselected by name: out of scope, needs `onnx`)
    assert "73 passed" in tail and "1 deselected" in tail and "15 subtests passed" in tail, tail
