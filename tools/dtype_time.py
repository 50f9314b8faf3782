import sys, os
sys.path.insert(0, os.getcwd())
import torch, numpy as np
import mct_quantizers_amd as mq
from mct_quantizers_amd.hip import native
Q = mq.pytorch_quantizers
def t(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for dt in (torch.float64, torch.float32, torch.float16):
    x = torch.randn(4096, 4096, device="cuda").to(dt)
    es = x.element_size()
    q = Q.WeightsSymmetricInferableQuantizer(8, [1.0 + 0.001 * i for i in range(4096)], True, 0)
    us = t(lambda: q(x)); print(dt, "per-channel axis0", round(us, 1), "us", round(2 * x.numel() * es / us / 1e6, 2), "TB/s", native.last_launch())
    q1 = Q.WeightsSymmetricInferableQuantizer(8, [1.0 + 0.001 * i for i in range(4096)], True, 1)
    us = t(lambda: q1(x)); print(dt, "per-channel axis1", round(us, 1), "us", round(2 * x.numel() * es / us / 1e6, 2), "TB/s", native.last_launch())
    qa = Q.ActivationUniformInferableQuantizer(8, [-2.5], [3.1])
    us = t(lambda: qa(x)); print(dt, "per-tensor", round(us, 1), "us", round(2 * x.numel() * es / us / 1e6, 2), "TB/s", native.last_launch())
    lut = [-128.0, -96.0, -64.0, -40.0, -24.0, -12.0, -5.0, 0.0, 5.0, 12.0, 24.0, 40.0, 64.0, 96.0, 120.0, 127.0]
    ql = Q.WeightsLUTSymmetricInferableQuantizer(4, lut, [1.0 + 0.001 * i for i in range(4096)], True, 0, 2)
    us = t(lambda: ql(x)); print(dt, "LUT per-channel", round(us, 1), "us", round(x.numel() * (es + 4) / us / 1e6, 2), "TB/s", native.last_launch())
