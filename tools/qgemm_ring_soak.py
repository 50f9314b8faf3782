"""Soak of the consumer's tiled kernel (ring / 8- / 16-wave variants and the automatic choice) on random RAGGED shapes
(M, N arbitrary, K any multiple of 16, also K smaller than a tile and smaller than the ring), both code types, with and
without bias, also with the K-rotation and stagger experiments switched on: every result against the exact integer product
(float64 matmul of the codes is exact here) and the float32 epilogue, bit for bit.   python tools/qgemm_ring_soak.py [seed]"""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mct_quantizers_amd.hip import native
lib = native.load(); dev = torch.device("cuda"); S = lambda: torch.cuda.current_stream().cuda_stream
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 11
g = torch.Generator(device=dev).manual_seed(seed); random.seed(seed)
VARIANTS = (0, 6623, 6624, 663, 664, 666, 6123, 6124, 12123, 12124, 12623, 12622, 12613, 12614, 3263, 3262, 6433, 3233,
            86623, 86622, 86633, 86433, 83233, 812613, 812123, 812122, 1612122, 1612623, 1612622, 166623)
bad = n = 0
for variant in VARIANTS:
    assert lib.mctq_set_tuning(b"ql_variant", variant) == 0
    for it in range(14):
        M = random.choice([1, 7, 16, 33, 64, 100, 129, 200, 256, 300, 513, 700])
        N = random.choice([16, 48, 100, 257, 512, 1000, 2048, 3000])
        K = 16 * random.choice([1, 2, 5, 8, 15, 16, 17, 31, 32, 33, 48, 64, 65, 100, 128, 200, 257])
        lib.mctq_set_tuning(b"ql_rot", it % 3 == 1); lib.mctq_set_tuning(b"ql_stagger", it % 4 == 2); lib.mctq_set_tuning(b"ql_band", random.choice([0, 0, 1, 3]))
        u8 = bool(random.getrandbits(1))
        a = torch.randint(0, 256, (M, K), dtype=torch.uint8, device=dev, generator=g) if u8 else torch.randint(-128, 128, (M, K), dtype=torch.int8, device=dev, generator=g)
        w = torch.randint(-128, 128, (N, K), dtype=torch.int8, device=dev, generator=g)
        za = random.randint(0, 255) if u8 else random.randint(-128, 127)
        sc = torch.rand(N, device=dev, generator=g) * 0.05 + 0.001
        bias = torch.randn(N, device=dev, generator=g) if random.getrandbits(1) else None
        rs = w.sum(1, dtype=torch.int32)
        frame = torch.full((M * N + 128,), 768.0, device=dev)
        y = frame[64:64 + M * N]
        rc = lib.mctq_qlinear_i8(a.data_ptr(), native.CODE_U8 if u8 else native.CODE_I8, za, 0.0173, w.data_ptr(), sc.data_ptr(), rs.data_ptr(),
                                 bias.data_ptr() if bias is not None else None, y.data_ptr(), M, N, K, S())
        assert rc == 0, lib.mctq_last_error()
        acc = ((a.double() - za) @ w.double().T).to(torch.int32).float() * (torch.tensor(0.0173, device=dev) * sc)
        want = (acc if bias is None else acc + bias).reshape(-1)
        n += 1
        if not torch.equal(y.view(torch.int32), want.view(torch.int32)) or not bool((frame[:64] == 768.0).all()) or not bool((frame[64 + M * N:] == 768.0).all()):
            bad += 1; print("MISMATCH", variant, M, N, K, u8, native.last_launch())
for k in (b"ql_variant", b"ql_rot", b"ql_stagger", b"ql_band"): lib.mctq_set_tuning(k, 0)
print("seed", seed, "checked", n, "mismatches", bad)
sys.exit(1 if bad else 0)
