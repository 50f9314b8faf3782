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
        "gathered_rows_match_local"}


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
