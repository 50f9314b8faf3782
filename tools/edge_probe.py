import sys, os
sys.path.insert(0, os.getcwd())
import torch, warnings
warnings.simplefilter("ignore")
import mct_quantizers_amd as mq
from mct_quantizers_amd.hip import ops
Q = mq.pytorch_quantizers
dev = "cuda"
def both(tag, f_ref, f_ours):
    out = []
    for f in (f_ref, f_ours):
        try:
            y = f()
            out.append(("ok", tuple(y.shape), y.stride(), y.dtype, y.flatten()[:3].tolist() if y.numel() else []))
        except Exception as e:
            out.append((type(e).__name__, str(e)[:90]))
    same = out[0] == out[1] if out[0][0] == "ok" else (out[0][0] == out[1][0] or issubclass(eval(out[1][0]) if out[1][0] in ("RuntimeError","NotImplementedError","ValueError","TypeError","AssertionError","IndexError") else Exception, RuntimeError) and out[0][0] == "RuntimeError")
    print(tag, "| aten:", out[0], "| ours:", out[1], "|", "SAME" if same else "DIFF", flush=True)
s, z = torch.tensor([0.1, 0.2, 0.3], device=dev), torch.zeros(3, dtype=torch.int32, device=dev)
x0 = torch.tensor(1.234, device=dev)
both("0-dim per-tensor", lambda: torch.fake_quantize_per_tensor_affine(x0, 0.1, 0, -128, 127), lambda: ops.fq_per_tensor(x0, 0.1, 0, -128, 127))
xe = torch.empty(0, 3, device=dev)
both("empty per-tensor", lambda: torch.fake_quantize_per_tensor_affine(xe, 0.1, 0, -128, 127), lambda: ops.fq_per_tensor(xe, 0.1, 0, -128, 127))
both("empty per-channel", lambda: torch.fake_quantize_per_channel_affine(xe, s, z, 1, -128, 127), lambda: ops.fq_per_channel(xe, s, z, 1, -128, 127))
x = torch.randn(4, 3, device=dev)
both("axis out of range", lambda: torch.fake_quantize_per_channel_affine(x, s, z, 2, -128, 127), lambda: ops.fq_per_channel(x, s, z, 2, -128, 127))
both("negative axis", lambda: torch.fake_quantize_per_channel_affine(x, s, z, -1, -128, 127), lambda: ops.fq_per_channel(x, s, z, -1, -128, 127))
both("scale length mismatch", lambda: torch.fake_quantize_per_channel_affine(x, s[:2], z[:2], 1, -128, 127), lambda: ops.fq_per_channel(x, s[:2], z[:2], 1, -128, 127))
xi = torch.ones(4, 3, dtype=torch.int32, device=dev)
both("int32 input per-tensor", lambda: torch.fake_quantize_per_tensor_affine(xi, 0.1, 0, -128, 127), lambda: ops.fq_per_tensor(xi, 0.1, 0, -128, 127))
both("int32 input per-channel", lambda: torch.fake_quantize_per_channel_affine(xi, s, z, 1, -128, 127), lambda: ops.fq_per_channel(xi, s, z, 1, -128, 127))
both("qmin>qmax per-tensor", lambda: torch.fake_quantize_per_tensor_affine(x, 0.1, 0, 5, 4), lambda: ops.fq_per_tensor(x, 0.1, 0, 5, 4))
both("zp out of range per-tensor", lambda: torch.fake_quantize_per_tensor_affine(x, 0.1, 300, -128, 127), lambda: ops.fq_per_tensor(x, 0.1, 300, -128, 127))
zbad = torch.tensor([0, 300, 0], dtype=torch.int32, device=dev)
both("zp out of range per-channel", lambda: torch.fake_quantize_per_channel_affine(x, s, zbad, 1, -128, 127), lambda: ops.fq_per_channel(x, s, zbad, 1, -128, 127))
both("float zp per-channel", lambda: torch.fake_quantize_per_channel_affine(x, s, z.float(), 1, -128, 127), lambda: ops.fq_per_channel(x, s, z.float(), 1, -128, 127))
both("double scales per-channel", lambda: torch.fake_quantize_per_channel_affine(x, s.double(), z, 1, -128, 127), lambda: ops.fq_per_channel(x, s.double(), z, 1, -128, 127))
both("scale 0 per-tensor", lambda: torch.fake_quantize_per_tensor_affine(x, 0.0, 0, -128, 127), lambda: ops.fq_per_tensor(x, 0.0, 0, -128, 127))
xh = x.half()
both("half per-channel", lambda: torch.fake_quantize_per_channel_affine(xh, s, z, 1, -128, 127), lambda: ops.fq_per_channel(xh, s, z, 1, -128, 127))
both("cpu scales gpu x", lambda: torch.fake_quantize_per_channel_affine(x, s.cpu(), z.cpu(), 1, -128, 127), lambda: ops.fq_per_channel(x, s.cpu(), z.cpu(), 1, -128, 127))
both("tqp 2-element scale", lambda: torch.fake_quantize_per_tensor_affine(x, s[:2], z[:2], -128, 127), lambda: ops.fq_per_tensor_tqp(x, s[:2], z[:2], -128, 127))
