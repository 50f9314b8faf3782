"""Why do the logits of the wrapped ResNet-50 differ in the last bits between the per-layer and the batched weight path
although every quantized weight is bit-equal?  Runs the same model twice in ONE mode, then both modes with a hook on every
module, and reports the first module whose output differs and whether its input and weight were bit-equal."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mct_quantizers_amd as mq
from mct_quantizers_amd import workloads

def run(model, x):
    rec = []
    hooks = []
    for name, m in model.named_modules():
        if isinstance(m, (mq.PytorchQuantizationWrapper, mq.PytorchActivationQuantizationHolder)):
            def hook(mod, inp, out, name=name):
                w = getattr(getattr(mod, "layer", None), "weight", None)
                rec.append((name, type(mod).__name__, inp[0].detach().clone(), None if w is None else w.detach().clone(), out.detach().clone()))
            hooks.append(m.register_forward_hook(hook))
    with torch.no_grad():
        y = model(x).clone()
    for h in hooks: h.remove()
    return y, rec

for batch, side in ((1, 224), (2, 64), (32, 224)):
    torch.manual_seed(0)
    x = torch.randn(batch, 3, side, side, device="cuda")
    model = workloads.wrapped_resnet50("cuda")
    y1, r1 = run(model, x)
    y2, r2 = run(model, x)
    print(f"batch {batch} @ {side}: per-layer twice: logits equal = {torch.equal(y1, y2)}", flush=True)
    mq.accelerate(model)
    with torch.no_grad(): model(x)
    y3, r3 = run(model, x)
    print(f"  per-layer vs batched weights: logits equal = {torch.equal(y1, y3)}, max abs diff {float((y1 - y3).abs().max()):.3e}")
    for (n, t, i1, w1, o1), (_, _, i3, w3, o3) in zip(r1, r3):
        same_in = torch.equal(i1, i3); same_w = w1 is None or torch.equal(w1, w3); same_out = torch.equal(o1, o3)
        if not same_out or not same_in or not same_w:
            print(f"  first difference at {n} ({t}): input equal {same_in}, weight equal {same_w}, output equal {same_out}; "
                  f"weight ptr alignment per-layer {0 if w1 is None else w1.data_ptr() % 4096}, max out diff {float((o1 - o3).abs().max()):.3e}")
            break
    else:
        print("  every wrapper / holder output equal")
    # the same layer called twice with identical tensors at different addresses
    conv = torch.nn.Conv2d(256, 64, 1, bias=False).cuda()
    xa = torch.randn(batch, 256, 56, 56, device="cuda")
    with torch.no_grad():
        ya = conv(xa)
        wb = conv.weight.detach().clone()
        pad = torch.empty(1237, device="cuda")            # shift the allocator
        wc = conv.weight.detach().clone()
        yb = torch.nn.functional.conv2d(xa, wb); yc = torch.nn.functional.conv2d(xa.clone(), wc)
    print(f"  plain conv2d, same values at other addresses: {torch.equal(ya, yb)} {torch.equal(ya, yc)}")
