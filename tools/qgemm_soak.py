import sys, os
sys.path.insert(0, os.getcwd())
import torch, random
from mct_quantizers_amd.hip import native
lib = native.load(); dev = torch.device("cuda"); S = lambda: torch.cuda.current_stream().cuda_stream
g = torch.Generator(device=dev).manual_seed(5); random.seed(5)
bad = 0; n = 0
for variant in (0, 2560, 2588, 2548, 2584, 2544):
    lib.mctq_set_tuning(b"ql_variant", variant)
    for _ in range(25):
        M = 128 * random.randint(1, 12); N = 128 * random.randint(1, 12); K = 128 * random.randint(2, 40)
        if variant == 0:
            M, N = 2048 + 256 * random.randint(0, 8), 4096 + 256 * random.randint(0, 8)
        u8 = bool(random.getrandbits(1))
        a = torch.randint(0, 256, (M, K), dtype=torch.uint8, device=dev, generator=g) if u8 else torch.randint(-128, 128, (M, K), dtype=torch.int8, device=dev, generator=g)
        w = torch.randint(-128, 128, (N, K), dtype=torch.int8, device=dev, generator=g)
        za = random.randint(0, 255) if u8 else random.randint(-128, 127)
        sc = torch.rand(N, device=dev, generator=g) * 0.05 + 0.001
        bias = torch.randn(N, device=dev, generator=g) if random.getrandbits(1) else None
        rs = w.sum(1, dtype=torch.int32)
        y = torch.empty(M, N, device=dev)
        rc = lib.mctq_qlinear_i8(a.data_ptr(), native.CODE_U8 if u8 else native.CODE_I8, za, 0.0173, w.data_ptr(), sc.data_ptr(), rs.data_ptr(), bias.data_ptr() if bias is not None else None, y.data_ptr(), M, N, K, S())
        if rc != 0:
            continue                      # shape refused by this tile variant
        acc = ((a.double() - za) @ w.double().T).to(torch.int32).float() * (torch.tensor(0.0173, device=dev) * sc)
        want = acc if bias is None else acc + bias
        n += 1
        if not torch.equal(y.view(torch.int32), want.view(torch.int32)):
            bad += 1; print("MISMATCH", variant, M, N, K, u8)
lib.mctq_set_tuning(b"ql_variant", 0)
print("checked", n, "mismatches", bad)
