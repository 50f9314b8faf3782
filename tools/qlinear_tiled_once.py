"""A few launches of the tiled integer consumer kernel (M = 2048, 4096 x 4096) for counter collection."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mct_quantizers_amd.hip import native
lib = native.load(); dev = torch.device("cuda")
M, N, K = 2048, 4096, 4096
a = torch.randint(0, 256, (M, K), dtype=torch.uint8, device=dev)
w = torch.randint(-128, 128, (N, K), dtype=torch.int8, device=dev)
sc = torch.rand(N, device=dev) * 0.01; rs = w.sum(1, dtype=torch.int32); y = torch.empty(M, N, device=dev)
for _ in range(5):
    lib.mctq_qlinear_i8(a.data_ptr(), native.CODE_U8, 114, 0.02, w.data_ptr(), sc.data_ptr(), rs.data_ptr(), None, y.data_ptr(), M, N, K,
                        torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
