"""bench.py's multi-process leg (bench_dist.py) on CPU: two processes, gloo backend.

No 8-GPU node is available to the build, so the code the driver runs at N = 2/4/8 -- process-group init,
barrier-bracketed timed region, max-over-ranks, config 5 sharded by dim 0 + one all-gather -- is executed here at
world size 2 on small shapes, and the keys of the emitted object are checked.
"""
import os
import socket

import torch
import torch.multiprocessing as mp

KEYS = {"workload", "scaling", "rows_per_rank", "compute_ms", "compute_elems_per_s", "allgather_ms",
        "allgather_recv_bytes_per_rank", "allgather_recv_gbs_per_rank", "compute_plus_allgather_elems_per_s",
        "gathered_rows_match_local", "ranks", "allgather_gbs_per_link", "allgather_output", "allgather_backend",
        "allgather_device", "allgather_local_copy_gbs", "gathered_equals_reference_digest", "allgather_staged_through_host",
        "ranks_shard_digest_ok"}


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, rows, q):
    import logging
    logging.getLogger("mct_quantizers_amd").setLevel(logging.ERROR)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import bench_dist
    device = torch.device("cpu")
    dist, control = bench_dist.init_process_group("gloo", device)
    try:
        assert control == "gloo" and dist.get_world_size() == world
        calls = []
        wall = bench_dist.timed_region(lambda i: calls.append(i), 7, device, dist)
        assert calls == list(range(7)) and wall > 0
        # the slowest rank defines the time: every rank gets the maximum
        got = bench_dist.max_over_ranks([float(rank + 1), 10.0 - rank], dist, control, device)
        assert got == [float(world), 10.0]
        leg = bench_dist.sharded_cfg5_leg(dist, rank, world, device, rows=rows, cols=128, reps=3, gather_reps=2)
        assert set(leg) == KEYS, set(leg) ^ KEYS
        assert leg["gathered_rows_match_local"] is True and leg["scaling"] == "strong"
        # every rank's shard against the REFERENCE's digest of that row block (tests/golden/shard_sha.json), gathered
        assert leg["ranks_shard_digest_ok"] == [True] * world and leg["allgather_staged_through_host"] is False
        if rank == 0:
            assert leg["gathered_equals_reference_digest"] is True
        per = -(-rows // world)
        assert leg["rows_per_rank"] == min(rows, (rank + 1) * per) - min(rows, rank * per)
        assert leg["allgather_recv_bytes_per_rank"] == (rows - leg["rows_per_rank"]) * 128 * 4
        assert leg["compute_elems_per_s"] > 0 and leg["compute_plus_allgather_elems_per_s"] > 0
        q.put((rank, leg["rows_per_rank"]))
    finally:
        dist.destroy_process_group()


def test_two_rank_bench_leg_emits_the_scale_keys():
    ctx = mp.get_context("spawn")
    for rows in (32, 13):                      # even split, and a short last block
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_worker, args=(r, 2, port, rows, q)) for r in range(2)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(timeout=240)
            assert p.exitcode == 0
        got = dict(q.get(timeout=10) for _ in range(2))
        assert got[0] + got[1] == rows


def test_watchdog_exits_nonzero():
    """A hung collective must not be reported as success: the watchdog prints the main line and exits 4."""
    import subprocess
    import sys
    code = ("import bench_dist, time\n"
            "with bench_dist.Watchdog(0.2, 0, lambda: 'MAIN-LINE'):\n"
            "    time.sleep(5)\n")
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=repo, capture_output=True, text=True, timeout=60)
    assert r.returncode == 4 and "MAIN-LINE" in r.stdout


def test_plain_python_bench_gpus_2_launches_itself_and_emits_the_multi_rank_line():
    """`python bench.py --gpus 2` with no RANK in the environment (how the driver launched BENCH in round 2) must start
    torch.distributed.run on itself as a child process and relay rank 0's JSON line and the exit status.  Here on CPU
    (--device cpu: the dry run of the same entry path over gloo -- launcher, process group, barriers, max-over-ranks,
    the sharded config-5 leg, the JSON)."""
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "5", "--warmup", "2", "--device", "cpu"],
                       cwd=repo, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 5 and out["warmup"] == 2 and out["value"] > 0
    assert out["ranks_seen"] == 2 and out["config"]["ranks_seen"] == 2 and out["config"]["control_plane"] == "gloo" and "dry_run" in out
    assert len(out["per_rank_kernel_us"]) == 2
    leg = out["sharded_cfg5"]
    assert leg["gathered_rows_match_local"] is True and leg["ranks"] == 2
    # self-verifying: per-rank digest of the last timed output (config 1 here) and of every rank's config-5 shard
    assert out["ranks_parity_ok"] == [True, True] and "parity_error" not in out
    assert leg["ranks_shard_digest_ok"] == [True, True] and leg["gathered_equals_reference_digest"] is True
    for key in ("sharded_cfg5_compute_elems_per_s", "sharded_cfg5_compute_plus_allgather_elems_per_s",
                "sharded_cfg5_allgather_gbs_per_link"):
        assert out[key] and out[key] > 0
    # a failing child is not reported as success
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "5", "--device", "cpu", "--config", "nope"],
                       cwd=repo, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra)
    return env


def test_no_multi_gpu_line_without_rccl_unless_allowed():
    """when the RCCL probe fails the rank EXITS with status 6 instead of continuing over gloo; only
    --allow-gloo (allow_gloo=True) turns the failure into a gloo control plane.  The probe failure is forced the way the
    GPU test forces it (MCTQ_BENCH_FORCE_GLOO -> force_gloo=True), so no GPU is needed to see both outcomes."""
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import torch, bench_dist\n"
            "d, c = bench_dist.init_process_group('nccl', torch.device('cpu'), force_gloo=True, allow_gloo=%s)\n"
            "print('CONTROL', c, d.get_world_size()); d.destroy_process_group()\n")
    env = _clean_env(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1")
    r = subprocess.run([sys.executable, "-c", code % "False"], cwd=repo, env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 6 and "CONTROL" not in r.stdout and "--allow-gloo not given" in r.stderr, (r.returncode, r.stderr[-500:])
    env["MASTER_PORT"] = str(_free_port())
    r = subprocess.run([sys.executable, "-c", code % "True"], cwd=repo, env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "CONTROL gloo 1" in r.stdout, (r.returncode, r.stderr[-500:])


def test_a_rank_whose_output_differs_from_the_reference_digest_fails_the_job(tmp_path):
    """the N > 1 line is self-verifying.  With a doctored digest for rank 1's config-5 shard the line
    says ranks_shard_digest_ok == [True, False] and the job exits 3; with a doctored digest of the timed workload every
    rank reports False (ranks_parity_ok) and the job exits 3 as well
    (the launcher relays that as a non-zero status of its own)."""
    import json
    import shutil
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    gold = os.path.join(repo, "tests", "golden")
    for which in ("shard", "full"):
        d = tmp_path / which
        d.mkdir()
        for name in ("full_sha.json", "shard_sha.json"):
            shutil.copy(os.path.join(gold, name), d / name)
        if which == "shard":
            doc = json.loads((d / "shard_sha.json").read_text())
            doc["cfg5"]["64x128"]["shards"]["2"][1] = "0" * 64
            (d / "shard_sha.json").write_text(json.dumps(doc))
        else:
            doc = json.loads((d / "full_sha.json").read_text())
            doc["configs"]["cfg1"]["y_sha256"] = "f" * 64
            (d / "full_sha.json").write_text(json.dumps(doc))
        r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--device", "cpu"],
                           cwd=repo, env=_clean_env(MCTQ_BENCH_GOLDEN_DIR=str(d)), capture_output=True, text=True, timeout=600)
        # rank 0 leaves with status 3; torch.distributed.run reports a failed child as its own status 1 and names the
        # child's status in its error table
        import re
        assert r.returncode != 0 and re.search(r"exitcode\s*:\s*3\b", r.stderr), (which, r.returncode, r.stderr[-1500:])
        out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
        assert "parity_error" in out
        if which == "shard":
            assert out["sharded_cfg5"]["ranks_shard_digest_ok"] == [True, False] and out["ranks_parity_ok"] == [True, True]
        else:
            assert out["ranks_parity_ok"] == [False, False] and out["sharded_cfg5"]["ranks_shard_digest_ok"] == [True, True]


def _bench(args, **env):
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    return subprocess.run([sys.executable, "bench.py", *args], cwd=repo, env=_clean_env(**env), capture_output=True, text=True, timeout=600)


def _line(r):
    import json
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, (r.stdout[-500:], r.stderr[-1500:])
    # ONE JSON line and nothing else on standard output: what libraries print to descriptor 1 on their own (RCCL's "Librccl path"
    # at exit, gloo's connection notes) goes to stderr (bench.py: own_stdout)
    assert [ln for ln in r.stdout.splitlines() if ln.strip()] == lines, r.stdout[-800:]
    return json.loads(lines[0])


def test_the_multi_rank_line_names_every_ranks_device_and_refuses_shared_ones():
    """the N > 1 line proves N devices.  Every rank reports the identity of the device it computes on
    (uuid | PCI address | name on a GPU; the process on the CPU dry run) -> ``rank_devices`` / ``devices_distinct``; two ranks
    on ONE device (forced here through MCTQ_BENCH_FAKE_DEVICE_ID) end the job with status 8 unless --allow-gloo marks the run as
    a rehearsal, and then the line says ``devices_distinct: false``."""
    import re
    base = ["--gpus", "2", "--steps", "3", "--warmup", "1", "--device", "cpu"]
    out = _line(_bench(base))
    assert len(out["rank_devices"]) == 2 and len(set(out["rank_devices"])) == 2 and out["devices_distinct"] is True
    assert all(d.startswith("cpu|pid ") for d in out["rank_devices"])
    r = _bench(base, MCTQ_BENCH_FAKE_DEVICE_ID="no-uuid|0000:00:00|a runtime that reports no identity")
    assert r.returncode == 0, r.stderr[-1500:]
    assert _line(r)["devices_distinct"] is None                      # unknown is not "shared": the job runs, the line says null
    r = _bench(base, MCTQ_BENCH_FAKE_DEVICE_ID="GPU-feed|0000:05:00|one device for both ranks")
    assert r.returncode != 0 and re.search(r"exitcode\s*:\s*8\b", r.stderr), (r.returncode, r.stderr[-1500:])
    assert "do not sit on 2 distinct devices" in r.stderr and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    r = _bench(base + ["--allow-gloo"], MCTQ_BENCH_FAKE_DEVICE_ID="GPU-feed|0000:05:00|one device for both ranks")
    assert r.returncode == 0, r.stderr[-1500:]
    out = _line(r)
    assert out["devices_distinct"] is False and out["rank_devices"] == ["GPU-feed|0000:05:00|one device for both ranks"] * 2


def test_the_single_rank_line_carries_the_three_sharded_series_too():
    """compute-only, compute + all-gather and per-link figures at EVERY N.  At N = 1 without a launcher bench.py
    forms a process group of one rank in-process and runs the same leg (per-link: null, no link is crossed); --no-sharded-extra
    skips it."""
    out = _line(_bench(["--steps", "3", "--warmup", "1", "--device", "cpu"]))
    assert out["n_gpus"] == 1 and out["rank_devices"] and out["devices_distinct"] is True
    leg = out["sharded_cfg5"]
    assert leg["ranks"] == 1 and leg["gathered_rows_match_local"] is True and leg["ranks_shard_digest_ok"] == [True]
    assert leg["gathered_equals_reference_digest"] is True and "one rank" in leg["process_group"]
    assert out["sharded_cfg5_compute_elems_per_s"] > 0 and out["sharded_cfg5_compute_plus_allgather_elems_per_s"] > 0
    assert "sharded_cfg5_allgather_gbs_per_link" in out and out["sharded_cfg5_allgather_gbs_per_link"] is None
    out = _line(_bench(["--steps", "3", "--warmup", "1", "--device", "cpu", "--no-sharded-extra"]))
    assert "sharded_cfg5" not in out


def test_device_identity_helpers():
    import bench_dist
    assert bench_dist.device_identity(torch.device("cpu")).startswith("cpu|pid ")
    assert bench_dist.gather_strings(None, "abc", None, torch.device("cpu")) == ["abc"]
    ids, distinct = bench_dist.rank_devices(None, None, torch.device("cpu"))
    assert len(ids) == 1 and distinct is True
    # a runtime without uuid and PCI address proves nothing either way: unknown, never a refusal
    assert bench_dist.identity_known("GPU-1234|0000:05:00|MI355X") and bench_dist.identity_known("no-uuid|0000:05:00|MI355X")
    assert not bench_dist.identity_known("no-uuid|0000:00:00|MI355X")
