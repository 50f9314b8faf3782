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
# every quantizer output inside the model: capture what each holder returns and what each wrapper installs
captured = {{}}
def _grab(name):
    def hook(mod, inp, out):
        if hasattr(mod, "activation_holder_quantizer"):
            captured[name + ".in"] = inp[0].detach().cpu().numpy(); captured[name + ".out"] = out.detach().cpu().numpy()
        else:
            for wname, _, _ in mod.get_weights_vars():
                captured[name + "." + wname] = getattr(mod.layer, wname).detach().cpu().numpy()
    return hook
for name, mod in model.named_children():
    if hasattr(mod, "activation_holder_quantizer") or hasattr(mod, "weights_quantizers"):
        mod.register_forward_hook(_grab(name))
y = model(x).detach().cpu().numpy()
np.save({out!r}, y)
np.savez({out!r} + ".captured.npz", **captured)
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


def _check_every_quantizer_bit_exact(tmp_path):
    """The quantizer outputs captured inside the un-pickled model, each recomputed by the ORACLE from the very input
    the quantizer saw on that device: bit equality per quantizer (only conv / linear / pooling around them may
    differ between devices)."""
    import pickle
    import warnings
    from oracle import oracle_call
    cap = np.load(str(tmp_path / "y.npy") + ".captured.npz")
    # constructor arguments of the pickled model (tools/gen_golden.py: gen_pickled_reference_models)
    holders = {"act": ("ActivationUniformInferableQuantizer", dict(num_bits=8, min_range=[-1.0], max_range=[3.0])),
               "fln": ("ActivationPOTInferableQuantizer", dict(num_bits=4, threshold=[2.0], signed=True)),
               "keep": ("ActivationLutPOTInferableQuantizer", dict(num_bits=2, lut_values=[-100.0, 0.0, 60.0, 127.0],
                                                                   threshold=[4.0], signed=True))}
    seen = 0
    for name, (cls, kw) in holders.items():
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            want = oracle_call(cls, kw, cap[name + ".in"])
        assert bits_equal(cap[name + ".out"], want), (name, first_mismatch(cap[name + ".out"], want, cap[name + ".in"]))
        seen += 1
    # weights: the float weights are in the pickle itself (the wrappers' parameters)
    import mct_quantizers_amd.compat as compat
    model = compat.load_reference_model(os.path.join(GOLDEN, "ref_model.pth"), map_location="cpu")
    conv_w = model.conv.weight.detach().numpy()
    thr_c = [float(v) for v in np.abs(conv_w).max(axis=(1, 2, 3))]
    checks = [("conv.weight", "WeightsSymmetricInferableQuantizer", dict(num_bits=8, threshold=thr_c, per_channel=True, channel_axis=0), conv_w),
              ("lin.weight", "WeightsLUTSymmetricInferableQuantizer",
               dict(num_bits=3, lut_values=[-100.0, -50.0, -10.0, 0.0, 10.0, 50.0, 100.0, 127.0], threshold=[1.0], per_channel=False),
               model.lin.weight.detach().numpy()),
              ("lin.bias", "WeightsUniformInferableQuantizer", dict(num_bits=8, min_range=[-1.0], max_range=[1.0], per_channel=False),
               model.lin.bias.detach().numpy())]
    for key, cls, kw, w in checks:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            want = oracle_call(cls, kw, w)
        assert bits_equal(cap[key], want), (key, first_mismatch(cap[key], want, w))
        seen += 1
    assert seen == 6


def test_reference_pickles_load_and_match_on_cpu(tmp_path):
    io = np.load(os.path.join(GOLDEN, "ref_model_io.npz"))
    y, yt = _run_child(tmp_path, "cpu")
    assert bits_equal(y, io["y"]), first_mismatch(y, io["y"])
    assert bits_equal(yt, io["y_traced"]), first_mismatch(yt, io["y_traced"])
    _check_every_quantizer_bit_exact(tmp_path)


@pytest.mark.gpu
def test_reference_pickles_run_on_the_hip_kernels(tmp_path):
    io = np.load(os.path.join(GOLDEN, "ref_model_io.npz"))
    y, yt = _run_child(tmp_path, "cuda")
    # every quantizer inside the model is held to bit equality with the oracle on the input it actually saw ...
    _check_every_quantizer_bit_exact(tmp_path)
    # ... only conv / linear / pooling around them run with the GPU's own summation order
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
