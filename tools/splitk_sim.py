import sys, os
sys.path.insert(0, os.getcwd())
import torch
from mct_quantizers_amd.hip import native
lib = native.load(); dev = torch.device("cuda"); S = lambda: torch.cuda.current_stream().cuda_stream
def t(M, N, K, variant):
    a = torch.randint(0, 256, (M, K), dtype=torch.uint8, device=dev)
    ring = max(2, int(400e6 // (N * K)) + 1)
    ws = [torch.randint(-128, 128, (N, K), dtype=torch.int8, device=dev) for _ in range(ring)]
    sc = torch.rand(N, device=dev) * 0.01; rs = ws[0].sum(1, dtype=torch.int32); b = torch.randn(N, device=dev)
    y = torch.empty(M, N, device=dev)
    lib.mctq_set_tuning(b"ql_variant", variant)
    call = lambda i: lib.mctq_qlinear_i8(a.data_ptr(), native.CODE_U8, 3, 0.02, ws[i % ring].data_ptr(), sc.data_ptr(), rs.data_ptr(), b.data_ptr(), y.data_ptr(), M, N, K, S())
    if call(0) != 0:
        return None
    for i in range(ring + 5): call(i)
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(100): call(i)
    e1.record(); torch.cuda.synchronize()
    lib.mctq_set_tuning(b"ql_variant", 0)
    return e0.elapsed_time(e1) * 10
for (M, N, K, v) in [(256, 4096, 4096, 0), (256, 16384, 1024, 2544), (256, 8192, 2048, 2544), (512, 4096, 4096, 0), (512, 16384, 1024, 2544), (512, 8192, 2048, 2544),
                     (1024, 4096, 4096, 0), (1024, 8192, 2048, 2544), (128, 4096, 4096, 0), (128, 16384, 1024, 2544), (128, 32768, 512, 2544)]:
    print(M, N, K, v, t(M, N, K, v), flush=True)
