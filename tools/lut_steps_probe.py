"""Wide integer codebooks (lut_values_bitwidth = 12 / 16): literal scan vs threshold list, per launch, on the cfg4 tensor
shape (4096 x 11008 float32, per channel).  Prints one line per (codebook size, kernel)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from mct_quantizers_amd.hip import native, ops


def timed(f, n=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    rng = np.random.default_rng(0)
    C, inner = 4096, 11008
    x = torch.randn(C, inner, device="cuda")
    thr = torch.from_numpy(rng.uniform(0.5, 3.0, C).astype(np.float32)).cuda()
    nbytes = x.numel() * 8
    for B, L in ((8, 16), (12, 16), (12, 64), (16, 256), (16, 1024)):
        lut = [float(v) for v in rng.choice(np.arange(-2 ** (B - 1), 2 ** (B - 1)), L, replace=False)]
        mult, cmin, cmax = float(2 ** (B - 1)), float(-2 ** (B - 1)), float(2 ** (B - 1) - 1)
        lut_d = torch.tensor(lut, device="cuda")
        tab = ops.make_lut_table(np.float32(lut), mult, cmin, cmax, "cuda")
        st = ops.make_lut_steps(np.float32(lut), mult, cmin, cmax, "cuda")
        y_lit = ops._hip_lut_per_channel(x, lut_d, thr, 1e-8, 0, mult, cmin, cmax, None)
        rows = [("literal", lambda: ops._hip_lut_per_channel(x, lut_d, thr, 1e-8, 0, mult, cmin, cmax, None))]
        if tab is not None:
            rows.append(("table", lambda: ops._hip_lut_per_channel(x, lut_d, thr, 1e-8, 0, mult, cmin, cmax, tab)))
        if st is not None:
            assert torch.equal(ops._hip_lut_per_channel(x, lut_d, thr, 1e-8, 0, mult, cmin, cmax, None, st), y_lit)
            rows.append(("steps", lambda: ops._hip_lut_per_channel(x, lut_d, thr, 1e-8, 0, mult, cmin, cmax, None, st)))
        for name, f in rows:
            us = timed(f)
            print(f"bitwidth={B:2d} L={L:4d} {name:8s} {us:9.1f} us  {nbytes / us / 1e6:7.2f} TB/s  [{native.last_launch()}]", flush=True)
        # per tensor too
        f_lit = lambda: ops._hip_lut_per_tensor(x, lut_d, 2.0, 2.0, mult, cmin, cmax, None)
        print(f"bitwidth={B:2d} L={L:4d} per-tensor literal {timed(f_lit):9.1f} us")
        if st is not None:
            f_st = lambda: ops._hip_lut_per_tensor(x, lut_d, 2.0, 2.0, mult, cmin, cmax, None, steps=st)
            assert torch.equal(f_st(), f_lit())
            print(f"bitwidth={B:2d} L={L:4d} per-tensor steps   {timed(f_st):9.1f} us")


if __name__ == "__main__":
    main()
