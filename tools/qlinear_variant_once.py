"""A few cold launches of mctq_qlinear_i8 at one problem size with one launch variant (counter collection):
python tools/qlinear_variant_once.py M VARIANT [N K]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mct_quantizers_amd.hip import native
lib = native.load(); dev = torch.device("cuda")
M, variant = int(sys.argv[1]), int(sys.argv[2])
N, K = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (4096, 4096)
ring = max(2, -(-400_000_000 // (N * K)))
ws = [torch.randint(-128, 128, (N, K), dtype=torch.int8, device=dev) for _ in range(ring)]
a = torch.randint(0, 256, (M, K), dtype=torch.uint8, device=dev)
sc = torch.rand(N, device=dev) * 0.01; rs = ws[0].sum(1, dtype=torch.int32); y = torch.empty(M, N, device=dev)
assert lib.mctq_set_tuning(b"ql_variant", variant) == 0
for i in range(ring + 6):
    assert lib.mctq_qlinear_i8(a.data_ptr(), native.CODE_U8, 114, 0.02, ws[i % ring].data_ptr(), sc.data_ptr(), rs.data_ptr(), None,
                               y.data_ptr(), M, N, K, torch.cuda.current_stream().cuda_stream) == 0
torch.cuda.synchronize()
print(native.last_launch())
