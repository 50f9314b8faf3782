"""Per-call host cost of the quantizer __call__ on small tensors (launch-bound regime)."""
import time, torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mct_quantizers_amd as mq
from mct_quantizers_amd.hip import native, ops
lib = native.load()
x = torch.randn(1, 3, 224, 224, device="cuda")
q = mq.pytorch_quantizers.ActivationUniformInferableQuantizer(8, [-2.5], [3.1])
qw = mq.pytorch_quantizers.WeightsSymmetricInferableQuantizer(8, [1.0] * 3, True, 1)
y = torch.empty_like(x)
st = torch.cuda.current_stream().cuda_stream
def bench(name, f, n=20000, calls_per_f=1):
    for _ in range(200): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    n *= calls_per_f
    print(f"{name:55s} host {1e6*(t1-t)/n:6.2f} us/call   incl. drain {1e6*(t2-t)/n:6.2f} us/call")
bench("quantizer(x)  ActivationUniform (per-tensor)", lambda: q(x))
bench("quantizer(x)  WeightsSymmetric per-channel axis1", lambda: qw(x))
fast = native.fast()
if fast is not None:
    plan = fast.AffinePlan(q.scale, q.zero_point, 0, 255)
    bench("compiled binding: AffinePlan(x)", lambda: plan(x))
    bench("compiled binding: fast.fq_per_tensor(x, ...)", lambda: fast.fq_per_tensor(x, q.scale, q.zero_point, 0, 255))
ql = mq.pytorch_quantizers.ActivationLutPOTInferableQuantizer(4, [-128.0, -96.0, -64.0, -40.0, -24.0, -12.0, -5.0, 0.0, 5.0, 12.0, 24.0, 40.0, 64.0, 96.0, 120.0, 127.0], [4.0], True)
bench("quantizer(x)  ActivationLutPOT (decision table)", lambda: ql(x))
holder = mq.PytorchActivationQuantizationHolder(q)
bench("PytorchActivationQuantizationHolder(q)(x)", lambda: holder(x))
bench("ops._hip_fq_per_tensor (ctypes binding)", lambda: ops._hip_fq_per_tensor(x, q.scale, q.zero_point, 0, 255))
bench("raw ctypes mctq_fq_per_tensor (no alloc)", lambda: lib.mctq_fq_per_tensor(x.data_ptr(), y.data_ptr(), x.numel(), 0, q.scale, q.zero_point, 0, 255, st))
bench("torch.empty_like", lambda: torch.empty_like(x))
bench("ATen torch.fake_quantize_per_tensor_affine (GPU)", lambda: torch.fake_quantize_per_tensor_affine(x, q.scale, q.zero_point, 0, 255))
s = torch.tensor([1.0] * 3, device="cuda"); z = torch.zeros(3, dtype=torch.int32, device="cuda")
bench("ATen torch.fake_quantize_per_channel_affine (GPU)", lambda: torch.fake_quantize_per_channel_affine(x, s, z, 1, -128, 127))
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(100): yy = q(x)
bench("hipGraph replay of 100 captured quantizer calls (per call)", lambda: g.replay(), n=200, calls_per_f=100)
