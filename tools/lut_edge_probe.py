import sys, os
sys.path.insert(0, os.getcwd())
import torch, warnings
warnings.simplefilter("ignore")
import mct_quantizers_amd as mq
from mct_quantizers_amd.hip import ops
Q = mq.pytorch_quantizers
lut = [-128.0, -64.0, -20.0, -5.0, 0.0, 5.0, 20.0, 64.0]
qa = Q.ActivationLutPOTInferableQuantizer(3, lut, [4.0], True)
lut_t = torch.tensor(lut)
def cmp(tag, x):
    try:
        want = ops._cpu_lut_per_tensor(x.cpu(), lut_t, 4.0 + 1e-8, 4.0, 128.0, -128.0, 127.0, -1)
        w = ("ok", tuple(want.shape), want.dtype, want.is_contiguous())
    except Exception as e:
        want = None; w = (type(e).__name__, str(e)[:60])
    try:
        got = qa(x)
        g = ("ok", tuple(got.shape), got.dtype, got.is_contiguous())
    except Exception as e:
        got = None; g = (type(e).__name__, str(e)[:60])
    eq = got is not None and want is not None and torch.equal(got.cpu(), want)
    print(tag, w, g, "values equal" if eq else "", flush=True)
cmp("0-dim", torch.tensor(1.3, device="cuda"))
cmp("empty", torch.empty(0, 5, device="cuda"))
cmp("1 elem", torch.tensor([[-7.0]], device="cuda"))
cmp("nan/inf", torch.tensor([float("nan"), float("inf"), -float("inf"), 0.0, -0.0], device="cuda"))
cmp("int32", torch.ones(3, dtype=torch.int32, device="cuda"))
cmp("f64", torch.randn(7, dtype=torch.float64, device="cuda"))
qw = Q.WeightsLUTSymmetricInferableQuantizer(3, lut, [1.0, 2.0, 0.5], True, 1, 2)
for tag, x in (("pc ok", torch.randn(4, 3, device="cuda")), ("pc empty", torch.empty(0, 3, device="cuda")), ("pc wrong C", torch.randn(4, 5, device="cuda")), ("pc rank", torch.randn(4, 3, 2, device="cuda"))):
    try:
        y = qw(x); print(tag, "ok", tuple(y.shape), y.dtype)
    except Exception as e:
        print(tag, type(e).__name__, str(e)[:80])
