"""Models pickled by the REFERENCE package load into this package's classes and give the reference's outputs.

tests/golden/ref_model.pth and ref_traced_holder.pth were written by tools/gen_golden.py with the
reference imported from /root/reference; they contain class paths and state, no source."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import GOLDEN, REPO, bits_equal, first_mismatch

_CHILD = r"""
import sys, numpy as np, torch, logging
sys.path.insert(0, {repo!r})
logging.getLogger("mct_quantizers_amd").setLevel(logging.ERROR)
assert "mct_quantizers" not in sys.modules
from mct_quantizers_amd import compat
import mct_quantizers_amd as mq
assert not compat.reference_is_installed()
io = np.load({io!r})
dev = {dev!r}
model = compat.load_reference_model({model!r}, map_location=dev)
assert "mct_quantizers" not in sys.modules                      # aliases are removed again
assert type(model.conv) is mq.PytorchQuantizationWrapper
assert type(model.fln) is mq.PytorchFLNActivationQuantizationHolder
assert type(model.lin.weights_quantizers["weight"]) is mq.pytorch_quantizers.WeightsLUTSymmetricInferableQuantizer
model = model.to(dev)
x = torch.from_numpy(io["x"]).to(dev)
y = model(x).detach().cpu().numpy()
np.save({out!r}, y)
traced = compat.load_reference_model({traced!r}, map_location=dev)
targets = [str(n.target) for n in traced.graph.nodes if n.op == "call_function"]
assert any("mctq_amd" in t for t in targets), targets             # the inlined ATen node was re-routed
np.save({out_t!r}, traced(x).detach().cpu().numpy())
"""


def _run_child(tmp_path, dev):
    out, out_t = str(tmp_path / "y.npy"), str(tmp_path / "yt.npy")
    code = _CHILD.format(repo=REPO, io=os.path.join(GOLDEN, "ref_model_io.npz"), dev=dev,
                         model=os.path.join(GOLDEN, "ref_model.pth"), traced=os.path.join(GOLDEN, "ref_traced_holder.pth"),
                         out=out, out_t=out_t)
    env = dict(os.environ)
    env["PYTHONPATH"] = ""                                          # the reference must NOT be importable
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    return np.load(out), np.load(out_t)


def test_reference_pickles_load_and_match_on_cpu(tmp_path):
    io = np.load(os.path.join(GOLDEN, "ref_model_io.npz"))
    y, yt = _run_child(tmp_path, "cpu")
    assert bits_equal(y, io["y"]), first_mismatch(y, io["y"])
    assert bits_equal(yt, io["y_traced"]), first_mismatch(yt, io["y_traced"])


@pytest.mark.gpu
def test_reference_pickles_run_on_the_hip_kernels(tmp_path):
    io = np.load(os.path.join(GOLDEN, "ref_model_io.npz"))
    y, yt = _run_child(tmp_path, "cuda")
    # the quantizers are bit-exact; conv/linear/pool around them run on the GPU with its own summation order
    assert np.allclose(y, io["y"], rtol=0, atol=2e-2), first_mismatch(y, io["y"])
    assert bits_equal(yt, io["y_traced"]), first_mismatch(yt, io["y_traced"])


def test_route_fx_graph_rewrites_inlined_aten_nodes():
    from mct_quantizers_amd import compat

    class M(torch.nn.Module):
        def forward(self, x, s, z):
            a = torch.fake_quantize_per_tensor_affine(x, scale=0.25, zero_point=3, quant_min=0, quant_max=15)
            b = torch.fake_quantize_per_channel_affine(a, s, z, 1, -8, 7)
            return b

    gm = torch.fx.symbolic_trace(M())
    x = torch.randn(2, 3, 5)
    s = torch.tensor([0.1, 0.2, 0.3])
    z = torch.zeros(3, dtype=torch.int32)
    want = gm(x, s, z)
    assert compat.route_fx_graph(gm) == 2
    assert torch.equal(gm(x, s, z), want)
    assert compat.route_fx_graph(gm) == 0


def test_aliases_refuse_to_shadow_a_real_install(monkeypatch):
    from mct_quantizers_amd import compat
    monkeypatch.setattr(compat, "reference_is_installed", lambda: True)
    with pytest.raises(RuntimeError):
        compat.install_reference_aliases()
