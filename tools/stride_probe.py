import sys, os
sys.path.insert(0, os.getcwd())
import torch, warnings
warnings.simplefilter("ignore")
import mct_quantizers_amd as mq
Q = mq.pytorch_quantizers
qa = Q.ActivationUniformInferableQuantizer(8, [-2.5], [3.1])
def show(tag, x):
    ref = torch.fake_quantize_per_tensor_affine(x, qa.scale, qa.zero_point, 0, 255)
    got = qa(x)
    print(tag, tuple(x.shape), x.stride(), "aten", ref.stride(), "ours", got.stride(), "same" if ref.stride() == got.stride() else "DIFF", torch.equal(ref, got))
base = torch.randn(8, 6, 5, 4, device="cuda")
show("contig", base)
show("perm", base.permute(0, 2, 3, 1))
show("perm-gap0", base.permute(0, 2, 3, 1)[::2])
show("perm-gap2", base.permute(0, 2, 3, 1)[:, :, ::2])
show("gap-last", base[..., ::2])
show("cl", base.contiguous(memory_format=torch.channels_last))
show("cl-gap", base.contiguous(memory_format=torch.channels_last)[::2])
show("t2d-gap", torch.randn(10, 12, device="cuda").t()[::2])
show("expand", torch.randn(1, 6, device="cuda").expand(4, 6))
qw = Q.WeightsSymmetricInferableQuantizer(8, [1.0] * 6, True, 1)
def showc(tag, x):
    ref = torch.fake_quantize_per_channel_affine(x, qw.scales, qw.zero_points, 1, -128, 127)
    got = qw(x.clone() if False else x)
    print("pc", tag, tuple(x.shape), x.stride(), "aten", ref.stride(), "ours", got.stride(), "same" if ref.stride() == got.stride() else "DIFF", torch.equal(ref, got))
b2 = torch.randn(8, 6, 5, 4, device="cuda")
showc("contig", b2); showc("cl", b2.contiguous(memory_format=torch.channels_last)); showc("cl-gap", b2.contiguous(memory_format=torch.channels_last)[::2])
showc("perm-gap", torch.randn(8, 5, 4, 6, device="cuda").permute(0, 3, 1, 2)[:, :, ::2])
