"""GPU: the multi-GPU leg as far as one GPU can run it -- bench.py under torch.distributed.run with RCCL at
world size 1 incl. the all_gather_into_tensor on device memory, the dim-0 row blocks of config 5 re-assembling to the
reference's digest, and the self-launching `python bench.py --gpus 2` entry path on two ranks sharing the device."""
import hashlib
import json
import os
import warnings

import numpy as np
import pytest
import torch

from conftest import GOLDEN, bits_equal, finite_equal, first_mismatch, load_json

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    from mct_quantizers_amd.hip import native
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    return native.load()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint64 if a.dtype == np.float64 else np.uint32)


def _make(cls, kwargs):
    import mct_quantizers_amd as mq
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return getattr(mq.pytorch_quantizers, cls)(**kwargs)


def test_bench_multi_process_leg_runs_over_rccl_on_one_gpu(lib):
    """bench.py launched as the driver launches it for N > 1 (torch.distributed.run, one process per GPU), here with one
    rank on the one GPU of the box: RCCL initialises, the barrier / max-over-ranks / sharded config-5 leg with its
    all_gather_into_tensor execute, and the JSON line carries the keys the scaling run reads."""
    import json
    import subprocess
    import sys
    from conftest import REPO
    import socket
    with socket.socket() as sk:                           # a port that is free right now (the rendezvous store binds it)
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(REPO, "bench.py"), "--gpus", "1", "--gather", "--steps", "20", "--warmup", "5",
           "--no-cpu", "--prewarm-seconds", "0.2", "--evidence-launches", "0"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=REPO)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-1000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["scaling"] == "weak" and d["value"] > 1e11
    assert d["config"]["control_plane"] == "nccl", d["config"]
    leg = d["sharded_cfg5"]
    for key in ("compute_ms", "compute_elems_per_s", "allgather_ms", "allgather_recv_bytes_per_rank", "compute_plus_allgather_elems_per_s", "gathered_rows_match_local"):
        assert key in leg, leg
    assert leg["gathered_rows_match_local"] is True and leg["rows_per_rank"] == 8192
    # the collective itself ran over RCCL on HIP memory (a group of one rank: the result is RCCL's copy of the shard, a
    # different buffer) and the re-assembled tensor is the reference's config-5 output
    assert leg["allgather_backend"] == "nccl" and leg["allgather_device"].startswith("cuda"), leg
    assert leg["allgather_ms"] > 0 and leg["allgather_local_copy_gbs"] > 100, leg
    assert leg["gathered_equals_reference_digest"] is True
    assert d["ranks_seen"] == 1


@pytest.mark.parametrize("world", [2, 8, 3])
def test_dim0_shards_reassemble_to_the_reference_digest_of_config_5(lib, world):
    """The multi-GPU partition of SURVEY 8(e) at FULL size on one GPU: each rank's row block of the 8192 x 8192 tensor,
    quantized with that rank's slice of the thresholds (sharded.shard_kwargs / row_block), concatenates to the tensor
    whose SHA-256 the reference produced (tests/golden/full_sha.json) -- even and ragged (3 ranks) splits."""
    import mct_quantizers_amd as mq
    from mct_quantizers_amd import sharded, workloads
    rec = load_json("full_sha.json")["configs"]["cfg5"]
    x_np = workloads.make_input("cfg5")
    wl = workloads.make_workload("cfg5", x_np)
    rows = x_np.shape[0]
    h = hashlib.sha256()
    covered = 0
    for rank in range(world):
        start, stop = sharded.row_block(rows, world, rank)
        q = getattr(mq.pytorch_quantizers, wl.quantizer)(**sharded.shard_kwargs(wl.kwargs, rows, world, rank))
        y = q(_dev(x_np[start:stop]))
        h.update(np.ascontiguousarray(y.cpu().numpy()).tobytes())
        covered += stop - start
    assert covered == rows and h.hexdigest() == rec["y_sha256"]


@pytest.mark.gpu
def test_plain_python_bench_gpus_2_on_the_gpu_box():
    """`python bench.py --gpus 2` as the driver launched BENCH in round 2 (no launcher, no RANK): the script starts
    torch.distributed.run on itself and rank 0 prints ONE line.  The box has one GPU, so both ranks share it
    (MCTQ_BENCH_WRAP_DEVICES) and the control plane is gloo (two RCCL ranks cannot share a device): everything of the
    N > 1 path except the RCCL collectives themselves runs on real kernels."""
    import json
    import os
    import subprocess
    import sys
    from conftest import REPO
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(MCTQ_BENCH_WRAP_DEVICES="1", MCTQ_BENCH_FORCE_GLOO="1")
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "20", "--warmup", "5", "--prewarm-seconds", "0.1",
                        "--evidence-launches", "0"], cwd=REPO, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["ranks_seen"] == 2 and d["config"]["control_plane"] == "gloo"
    assert len(d["per_rank_kernel_us"]) == 2 and all(v > 5 for v in d["per_rank_kernel_us"])
    # (two processes time-slicing ONE GPU: the rate is not a measurement, only sanity)
    assert d["value"] > 1e9 and d["scaling"] == "weak" and "rows_kernel" in d["roofline"]["kernel"]
    assert d["ranks_seen"] == 2
