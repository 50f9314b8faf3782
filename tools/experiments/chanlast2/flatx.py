#!/usr/bin/env python3
"""flat_x (chanlast2.hip): the per-tensor launch with something between the arrival of the data and the arithmetic -- a dependent table
read (a different word per block / one hot word), an s_sleep, the read issued before the data loads -- beside the shipped per-tensor
launch (flat_kernel) and the shipped short-row launch with equal scales (the same output bits).  Arms timed alternately on the same buffers.
    python tools/build_variant.py chanlast2 && python tools/experiments/chanlast2/flatx.py [bf16|f16|f32] [rows] [cols]"""
import ctypes
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, REPO)
import torch

xlib = ctypes.CDLL(os.path.join(REPO, "tools", "ablate", "libmctq_hip_chanlast2.so"))
P, I64, I32, F32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_float
xlib.mctq_fq_per_channel.argtypes = [P, P, I64, I64, I64, I32, P, P, I32, I32, P]
xlib.mctq_fq_per_tensor.argtypes = [P, P, I64, I32, F32, I32, I32, I32, P]
xlib.mctq_x_flat.argtypes = [I32, I32, P, P, I64, I32, F32, I32, I32, I32, P, P]
xlib.mctq_last_launch.restype = ctypes.c_char_p
dt_name = sys.argv[1] if len(sys.argv) > 1 else "bf16"
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
cols = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
tdt, dtc, esz = {"f32": (torch.float32, 0, 4), "f16": (torch.float16, 1, 2), "bf16": (torch.bfloat16, 2, 2)}[dt_name]
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
n = rows * cols
nb = n * esz * 2
nt = 2 if nb // 2 <= (32 << 20) else 1
ring = min(64, max(2, -(-(512 << 20) // nb) + 1))
xs = [(torch.randn(rows, cols, device=dev) * 2).to(tdt) for _ in range(ring)]
ys = [torch.empty_like(x) for x in xs]
S = 0.03
table = torch.full((4096,), S, device=dev)
s_same = torch.full((rows,), S, device=dev)


def lib_flat(i):
    assert xlib.mctq_fq_per_tensor(xs[i % ring].data_ptr(), ys[i % ring].data_ptr(), n, dtc, S, 0, -128, 127, stream) == 0


def lib_short(i):
    assert xlib.mctq_fq_per_channel(xs[i % ring].data_ptr(), ys[i % ring].data_ptr(), 1, rows, cols, dtc, s_same.data_ptr(), None, -128, 127, stream) == 0


def fx(mode, sleep=0):
    def call(i):
        assert xlib.mctq_x_flat(mode, sleep, xs[i % ring].data_ptr(), ys[i % ring].data_ptr(), n, dtc, S, -128, 127, nt, table.data_ptr(), stream) == 0
    return call


if os.environ.get("FLATX_ARMS") == "explicit":
    arms = [("shipped per-tensor (flat_kernel)", lib_flat), ("shipped per-channel, equal scales", lib_short), ("flat_x 0: masked, compiler's waits", fx(0)),
            ("flat_x 8: wait all + PACED", fx(8)), ("flat_x 10: 8 + loads 64 clk apart", fx(10)), ("flat_x 11: 8 + masked loads", fx(11)),
            ("flat_x 13: 8 + masked arithmetic", fx(13)), ("flat_x 5: PACED only", fx(5))]
elif os.environ.get("FLATX_ARMS") == "spaced":
    arms = [("shipped per-tensor (flat_kernel)", lib_flat), ("shipped per-channel, equal scales", lib_short),
            ("flat_x 10: spaced 64 clk + all first + PACED", fx(10, 1)), ("flat_x 10: spaced 128 clk", fx(10, 2)), ("flat_x 10: spaced 256 clk", fx(10, 4)),
            ("flat_x 10: spaced 512 clk", fx(10, 8)), ("flat_x 14: spaced loads ONLY (64 clk)", fx(14, 1)), ("flat_x 14: spaced loads ONLY (256 clk)", fx(14, 4)),
            ("flat_x 15: spaced + all loads first", fx(15)), ("flat_x 16: spaced + PACED", fx(16)), ("flat_x 6: the shipped shape", fx(6))]
else:
  arms = [("shipped per-tensor (flat_kernel)", lib_flat), ("shipped per-channel, equal scales", lib_short), ("flat_x 0: nothing", fx(0)),
        ("flat_x 1: table word per block", fx(1)), ("flat_x 2: one hot table word", fx(2)), ("flat_x 4: word per block, read first", fx(4)),
        ("flat_x 5: full tile, PACED stores", fx(5)), ("flat_x 6: full tile, unpaced", fx(6)), ("flat_x 7: paced from store 2", fx(7)), ("flat_x 8: full tile, wait all + PACED", fx(8)), ("flat_x 9: full tile, wait all, unpaced", fx(9)),
        ("flat_x 3: s_sleep 4", fx(3, 4)),]
view = torch.int16 if esz == 2 else torch.int32
want = None
for name, call in arms:
    ys[0].zero_()
    call(0)
    torch.cuda.synchronize()
    if want is None:
        want = ys[0].clone()
    assert torch.equal(ys[0].view(view), want.view(view)), name
for name, call in arms:
    for i in range(2000):
        call(i)
torch.cuda.synchronize()
times = {name: [] for name, _ in arms}
for rnd in range(5):
    for name, call in arms:
        for i in range(50):
            call(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(200):
            call(i)
        e1.record()
        torch.cuda.synchronize()
        times[name].append(e0.elapsed_time(e1) * 1e3 / 200)
print(f"{dt_name} {rows}x{cols}, {nb >> 20} MiB per launch, ring {ring}, store policy NT={nt}; all arms bit-equal")
for name, _ in arms:
    t = sorted(times[name])
    print(f"  {name:40s} {t[0]:6.2f} / {t[2]:6.2f} / {t[-1]:6.2f} us (min / median / max of 5)  frac {nb / t[2] / 8e6:.3f}")
