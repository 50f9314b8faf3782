"""Tile size of the batched kernel: the library built with -DMCTQ_BATCH_U=2 / 4 / 8 (tools/ablate/libmctq_U*.so, loaded
through MCTQ_HIP_LIB with the ctypes binding), 16 x 4096^2 and ResNet-50's weights through mctq_fq_batch_pack / _run."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MCTQ_BINDING"] = "ctypes"
import numpy as np
import torch
from mct_quantizers_amd.hip import native
from requant_model_weights import resnet50_shapes

lib = native.load()
st = torch.cuda.current_stream().cuda_stream


def run(name, shapes, sets, reps):
    plans = []
    nbytes = 0
    for s in range(sets):
        arr = (native.FqItem * len(shapes))()
        keep = []
        for it, sh in zip(arr, shapes):
            x = torch.randn(sh, device="cuda") * 0.05
            y = torch.empty_like(x)
            sc = (torch.rand(sh[0], device="cuda") * 0.01 + 0.001)
            inner = int(np.prod(sh[1:]))
            it.x, it.y, it.outer, it.channels, it.inner = x.data_ptr(), y.data_ptr(), 1, sh[0], inner
            it.scales, it.zero_points, it.quant_min, it.quant_max, it.dtype, it.flags = sc.data_ptr(), None, -128, 127, native.DT_F32, 0
            keep.append((x, y, sc))
            if s == 0:
                nbytes += x.numel() * 8
        need = lib.mctq_fq_batch_pack(arr, len(shapes), None, 0)
        host = np.zeros(need, np.uint8)
        assert lib.mctq_fq_batch_pack(arr, len(shapes), host.ctypes.data, need) == need
        dev = torch.from_numpy(host).cuda()
        plans.append((host, dev, keep))
    call = lambda i: lib.mctq_fq_batch_run(plans[i % sets][0].ctypes.data, plans[i % sets][1].data_ptr(), st)
    for i in range(sets + 3):
        assert call(i) == 0, lib.mctq_last_error()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        call(i)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    x, y, sc = plans[0][2][0]
    z = torch.zeros(sc.numel(), dtype=torch.int32, device="cuda")
    ok = torch.equal(y, torch.fake_quantize_per_channel_affine(x, sc, z, 0, -128, 127))
    print(f"{os.environ.get('MCTQ_HIP_LIB', 'default'):40s} {name:28s} {us:8.1f} us {nbytes / us / 1e3:6.0f} GB/s  ok={ok} [{native.last_launch()}]", flush=True)


run("16 x 4096^2", [(4096, 4096)] * 16, 2, 30)
run("4 x 8192^2", [(8192, 8192)] * 4, 2, 30)
run("ResNet-50", resnet50_shapes(), 4, 100)
