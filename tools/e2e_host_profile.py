"""Where does the HOST time of an eager forward of the wrapped ResNet-50 go?  cProfile of N forwards per mode
(per-layer / auto-batched), plus direct timings of the weight re-quantization alone in both forms."""
import cProfile, io, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mct_quantizers_amd as mq
from mct_quantizers_amd import workloads
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1
x = torch.randn(batch, 3, 224, 224, device="cuda")
def wall(fn, n=100):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    h = time.perf_counter() - t; torch.cuda.synchronize(); return h / n * 1e6, (time.perf_counter() - t) / n * 1e6
for mode in ("per_layer", "auto_batched"):
    model = workloads.wrapped_resnet50("cuda")
    if mode == "auto_batched": mq.accelerate(model)
    with torch.no_grad():
        for _ in range(5): model(x)
        host, total = wall(lambda: model(x), 60)
        print(f"== {mode}: forward host {host:.0f} us, incl. drain {total:.0f} us", flush=True)
        wrappers = [m for m in model.modules() if isinstance(m, mq.PytorchQuantizationWrapper)]
        if mode == "per_layer":
            h, t = wall(lambda: [q(w) for m in wrappers for _, w, q in m.get_weights_vars()])
            print(f"   54 quantizer calls alone: host {h:.0f} us, incl. drain {t:.0f} us")
            layers = [(m.layer, x) for m in wrappers]
        else:
            hnd = mq.accelerated(model)
            h, t = wall(hnd.quantize_now)
            print(f"   handle.quantize_now(): host {h:.0f} us, incl. drain {t:.0f} us")
        # the same forward with the quantizer work removed entirely (weights as they are): what is left is torch + MIOpen
        saved = [(m, m._weights_vars) for m in wrappers]
        for m in wrappers: m._weights_vars = []
        h0, t0 = wall(lambda: model(x), 60)
        for m, v in saved: m._weights_vars = v
        print(f"   forward WITHOUT any weight re-quantization: host {h0:.0f} us, incl. drain {t0:.0f} us", flush=True)
        pr = cProfile.Profile(); pr.enable()
        for _ in range(30): model(x)
        torch.cuda.synchronize(); pr.disable()
        s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(14)
        print("\n".join(l[:150] for l in s.getvalue().splitlines()[4:30]))
