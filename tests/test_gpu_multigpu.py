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
    assert leg["gathered_equals_reference_digest"] is True and leg["allgather_staged_through_host"] is False
    assert d["ranks_seen"] == 1
    # self-verifying line: the rank's last timed output and its (whole-tensor) shard against the reference's digests
    assert d["ranks_parity_ok"] == [True] and leg["ranks_shard_digest_ok"] == [True] and "parity_error" not in d


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


def _bench_wrapped(n, *flags, env_extra=None, timeout=1500, wrap_env=True):
    """`python bench.py --gpus n` with no launcher and no RANK, all ranks sharing the box's one GPU."""
    import subprocess
    import sys
    from conftest import REPO
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.pop("MCTQ_BENCH_WRAP_DEVICES", None)
    env.update(**({"MCTQ_BENCH_WRAP_DEVICES": "1"} if wrap_env else {}), **(env_extra or {}))
    return subprocess.run([sys.executable, "bench.py", "--gpus", str(n), "--steps", "20", "--warmup", "5", "--prewarm-seconds", "0.1",
                           "--evidence-launches", "0", *flags], cwd=REPO, env=env, capture_output=True, text=True, timeout=timeout)


def _check_self_verifying_line(d, n):
    assert d["n_gpus"] == n and d["ranks_seen"] == n and d["config"]["ranks_seen"] == n and d["config"]["control_plane"] == "gloo"
    assert len(d["per_rank_kernel_us"]) == n and all(v > 5 for v in d["per_rank_kernel_us"])
    # (n processes time-slicing ONE GPU: the rate is not a measurement, only sanity)
    assert d["value"] > 1e9 and d["scaling"] == "weak" and "rows_kernel" in d["roofline"]["kernel"]
    # every rank compared its own last timed output with the reference's config-2 digest ...
    assert d["ranks_parity_ok"] == [True] * n and "parity_error" not in d
    # ... and its own config-5 shard with the reference's digest of that row block; rank 0 the re-assembled tensor
    leg = d["sharded_cfg5"]
    assert "error" not in leg, leg
    assert leg["ranks"] == n and leg["ranks_shard_digest_ok"] == [True] * n and leg["rows_per_rank"] == 8192 // n
    assert leg["gathered_rows_match_local"] is True and leg["gathered_equals_reference_digest"] is True
    assert leg["allgather_staged_through_host"] is True and leg["allgather_backend"] == "gloo"
    # rank 0 ran the CPU leg after the timed region at N > 1 too
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 1e7 and cb["gpu_output_bit_equal"] is True


def test_plain_python_bench_gpus_2_on_the_gpu_box():
    """`python bench.py --gpus 2` as the driver launched BENCH in round 2 (no launcher, no RANK): the script starts
    torch.distributed.run on itself and rank 0 prints ONE line.  The box has one GPU, so both ranks share it
    (MCTQ_BENCH_WRAP_DEVICES); two RCCL ranks cannot share a device, so WITHOUT --allow-gloo the job must refuse to print a
    line (status 6 in the rank, non-zero from the launcher), and WITH it everything of the N > 1 path except the RCCL
    collectives themselves runs on real kernels: gloo control plane, per-rank digests, host-staged gather, CPU leg."""
    import re
    r = _bench_wrapped(2, "--cpu-seconds", "2", env_extra={"MCTQ_BENCH_FORCE_GLOO": "1"})
    assert r.returncode != 0 and re.search(r"exitcode\s*:\s*6\b", r.stderr), (r.returncode, r.stderr[-2000:])
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")], r.stdout[-500:]
    r = _bench_wrapped(2, "--cpu-seconds", "2", "--allow-gloo", env_extra={"MCTQ_BENCH_FORCE_GLOO": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    _check_self_verifying_line(json.loads(lines[0]), 2)


def test_local_rank_beyond_the_visible_devices_is_judged_by_identity():
    """A launcher that hands every rank its own HIP_VISIBLE_DEVICES leaves LOCAL_RANK >= device_count(): the rank then takes
    device LOCAL_RANK % count and the exchanged identities decide.  Here (one GPU, no MCTQ_BENCH_WRAP_DEVICES) both ranks
    land on the same device: the line says so (devices_distinct false, two equal rank_devices) instead of the job dying on
    an invalid device ordinal."""
    r = _bench_wrapped(2, "--cpu-seconds", "1", "--allow-gloo", "--no-sharded-extra", env_extra={"MCTQ_BENCH_FORCE_GLOO": "1"},
                       wrap_env=False)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["devices_distinct"] is False and len(d["rank_devices"]) == 2 and d["rank_devices"][0] == d["rank_devices"][1]
    assert d["ranks_parity_ok"] == [True, True]


def test_eight_ranks_rehearsed_on_the_one_gpu():
    """the command the driver's SCALE run issues, `python bench.py --gpus 8`, with the eight ranks
    wrapped onto the one GPU of the box (RCCL's own probe fails here -- eight ranks, one device -- so --allow-gloo): the
    line carries ranks_parity_ok / ranks_shard_digest_ok for all eight ranks, 1024-row shards re-assembling to the
    reference's digest, and rank 0's cpu_baseline."""
    r = _bench_wrapped(8, "--cpu-seconds", "2", "--allow-gloo")
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    _check_self_verifying_line(json.loads(lines[0]), 8)
