"""Pointwise (1x1) convolution blocks: fake-quant kernels + float32 convolution vs the integer consumer."""
import sys, os, time, warnings, logging
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
logging.getLogger("mct_quantizers_amd").setLevel(logging.ERROR)
warnings.filterwarnings("ignore")
import torch
import mct_quantizers_amd as mq
from mct_quantizers_amd import consumers
Q = mq.pytorch_quantizers
def block(ci, co):
    torch.manual_seed(0)                     # the two copies must share weights AND the thresholds derived from them
    conv = torch.nn.Conv2d(ci, co, 1).cuda()
    thr = [float(v) for v in conv.weight.detach().abs().amax(dim=(1, 2, 3))]
    return torch.nn.Sequential(mq.PytorchActivationQuantizationHolder(Q.ActivationUniformInferableQuantizer(8, [-3.0], [3.0])),
                               mq.PytorchQuantizationWrapper(conv, {"weight": Q.WeightsSymmetricInferableQuantizer(8, thr, True, 0)}))
def timeit(m, x, n=50):
    with torch.no_grad():
        for _ in range(5): y = m(x)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(n): y = m(x)
        torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e6, y
for (b, ci, co, hw) in [(32, 96, 576, 56), (32, 576, 160, 14), (8, 256, 1024, 28), (64, 64, 256, 56)]:
    for fmt in (torch.channels_last, torch.contiguous_format):
        x = torch.randn(b, ci, hw, hw, device="cuda").contiguous(memory_format=fmt)
        ref = block(ci, co); fused = block(ci, co)
        consumers.fuse_linear_consumers(fused)
        t0, y0 = timeit(ref, x); t1, y1 = timeit(fused, x)
        print(f"[{b},{ci},{hw},{hw}] -> {co} {'NHWC' if fmt is torch.channels_last else 'NCHW'}: fake-quant + fp32 conv {t0:8.1f} us, "
              f"integer consumer {t1:8.1f} us (x{t0 / t1:.2f}), max rel diff {float((y1 - y0).abs().max() / y0.abs().max()):.1e}", flush=True)
