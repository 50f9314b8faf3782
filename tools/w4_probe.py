import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mct_quantizers_amd.hip import native
from mct_quantizers_amd import consumers
lib = native.load(); dev = torch.device("cuda"); S = lambda: torch.cuda.current_stream().cuda_stream
for (M, N, K) in [(1, 8192, 8192), (16, 8192, 8192), (16, 28672, 8192), (16, 8192, 28672), (32, 8192, 8192), (64, 8192, 8192)]:
    ring = max(2, int(np.ceil(600e6 / (N * K))))
    w8 = [torch.randint(-8, 8, (N, K), dtype=torch.int8, device=dev) for _ in range(ring)]
    w4 = [consumers.pack_w4(w) for w in w8]
    a = torch.randint(0, 256, (M, K), dtype=torch.uint8, device=dev)
    sc = torch.rand(N, device=dev) * 0.01; wsum = w8[0].sum(1, dtype=torch.int32); bias = torch.randn(N, device=dev)
    y = torch.empty(M, N, dtype=torch.float32, device=dev)
    def t8(i): lib.mctq_qlinear_i8(a.data_ptr(), native.CODE_U8, 114, 0.02, w8[i % ring].data_ptr(), sc.data_ptr(), wsum.data_ptr(), bias.data_ptr(), y.data_ptr(), M, N, K, S())
    def t4(i): lib.mctq_qlinear_w4a8(a.data_ptr(), native.CODE_U8, 114, 0.02, w4[i % ring].data_ptr(), sc.data_ptr(), wsum.data_ptr(), bias.data_ptr(), y.data_ptr(), -1, 1.0, 0, 0, 0, M, N, K, S())
    for name, f in (("int8", t8), ("w4", t4)):
        for i in range(5): f(i)
        torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(40): f(i)
        e1.record(); torch.cuda.synchronize()
        print(M, N, K, name, round(e0.elapsed_time(e1) * 1000 / 40, 2), "us", flush=True)
    del w8, w4
