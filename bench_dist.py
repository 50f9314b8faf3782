"""Importable pieces of bench.py's multi-process leg (one process per GPU; RCCL = backend "nccl" on ROCm).

They take the device and the backend as arguments so that the very code the driver runs on 2/4/8 GPUs is
exercised by a world-size-2 ``gloo`` test on CPU (tests/test_bench_dist_gloo.py) and by
``torchrun --nproc-per-node 1 bench.py --gpus 1 --gather`` on the single GPU of a gpurun box.
No 8-GPU node is available to the build: the N > 1 numbers themselves come from the driver's SCALE run.
"""
from __future__ import annotations

import os
import sys
import time
from typing import Callable, Dict, Optional, Tuple

import torch


def free_port() -> int:
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(script: str, n: int, argv) -> int:
    """``python bench.py --gpus N`` without a launcher: start ``python -m torch.distributed.run`` (one rank per GPU) on
    the same script and arguments as a CHILD process and return its exit status.  Called before anything has touched
    the GPU (a process that has initialised HIP must not be replaced, and is not: the parent only waits).  The child's
    stdout / stderr are inherited, so rank 0's JSON line is this process's output."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), script] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL needs on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    print("[bench] no RANK in the environment: launching " + " ".join(cmd[1:8]) + " ...", file=sys.stderr, flush=True)
    return subprocess.run(cmd, env=env).returncode


EXIT_NO_RCCL = 6          # --gpus N > 1 without a working RCCL group and without --allow-gloo
EXIT_NO_SHARDED_LEG = 7   # --gpus N > 1 whose sharded config-5 + all-gather leg did not run (or failed) without --allow-gloo
EXIT_SHARED_DEVICE = 8    # --gpus N > 1 whose ranks do not sit on N distinct devices (or MCTQ_BENCH_WRAP_DEVICES is set) without --allow-gloo


def device_identity(device: torch.device) -> str:
    """What tells one physical device from another in a bench line: ``<uuid>|<pci domain:bus:device>|<name>`` of the HIP
    device this rank computes on (``torch.cuda.get_device_properties``); for the CPU dry run the process itself
    (``cpu|pid``).  MCTQ_BENCH_FAKE_DEVICE_ID replaces it (tests: two ranks that claim the same device)."""
    fake = os.environ.get("MCTQ_BENCH_FAKE_DEVICE_ID")
    if fake:
        return fake
    if device.type != "cuda":
        return f"cpu|pid {os.getpid()}"
    p = torch.cuda.get_device_properties(device)
    uuid = str(getattr(p, "uuid", "") or "")
    if not uuid.strip("0-") or uuid.lower() in ("none", "no-uuid"):
        uuid = "no-uuid"
    bus = "%04x:%02x:%02x" % (int(getattr(p, "pci_domain_id", 0)), int(getattr(p, "pci_bus_id", 0)), int(getattr(p, "pci_device_id", 0)))
    return f"{uuid}|{bus}|{p.name}"


def identity_known(ident: str) -> bool:
    """Does the string tell one device from another?  (A runtime that reports neither a uuid nor a PCI address leaves
    ``no-uuid|0000:00:00|<name>``: equal strings then prove nothing, and the job must not be refused on them.)"""
    parts = ident.split("|")
    if len(parts) < 2 or parts[0] == "cpu":
        return True
    return parts[0] != "no-uuid" or parts[1] != "0000:00:00"


def gather_strings(dist, text: str, control: Optional[str], device: torch.device, width: int = 256):
    """[text of rank 0, text of rank 1, ...] on every rank (fixed-width byte tensors: works on RCCL and on gloo alike)."""
    if dist is None:
        return [text]
    raw = text.encode("utf-8", "replace")[:width]
    t = torch.zeros(width, dtype=torch.uint8)
    t[:len(raw)] = torch.tensor(list(raw), dtype=torch.uint8)
    t = t.to(device if control == "nccl" else "cpu")
    got = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(got, t)
    return [bytes(v.cpu().tolist()).rstrip(b"\0").decode("utf-8", "replace") for v in got]


def rank_devices(dist, control: Optional[str], device: torch.device):
    """(identities of every rank's device, are they pairwise distinct: True / False, or None when the runtime gives no identity)."""
    ids = gather_strings(dist, device_identity(device), control, device)
    if not all(identity_known(i) for i in ids):
        return ids, None
    return ids, len(set(ids)) == len(ids)


def init_single_rank_group(device: torch.device):
    """A process group of ONE rank formed in-process (no launcher): lets the N = 1 line run the same sharded config-5 +
    all-gather leg the N > 1 lines run, so the three series of SURVEY 8(e) exist at every N.  RCCL on a GPU, gloo on the
    CPU dry run.  Returns (dist, control) or raises."""
    import torch.distributed as dist
    method = f"tcp://127.0.0.1:{free_port()}"
    if device.type == "cuda":
        dist.init_process_group("nccl", init_method=method, rank=0, world_size=1, device_id=device)
        probe = torch.zeros(1, device=device)
        dist.all_reduce(probe)
        torch.cuda.synchronize(device)
        return dist, "nccl"
    dist.init_process_group("gloo", init_method=method, rank=0, world_size=1)
    return dist, "gloo"


def init_process_group(backend: str, device: torch.device, force_gloo: bool = False, allow_gloo: bool = False):
    """Returns (dist module, control_plane).  ``backend`` "nccl" is probed with one all-reduce (RCCL creates its
    communicators lazily: fail here, not mid-run).  If the probe fails the process EXITS with status EXIT_NO_RCCL: a
    multi-GPU line must not be printed by a job that has no RCCL group (a SCALE run could otherwise come
    back rc 0 with no collective in it).  ``allow_gloo`` (bench.py --allow-gloo: rehearsals of the N > 1 entry path on a
    box with fewer GPUs than ranks) replaces RCCL by gloo for the CONTROL plane instead (barrier, max-over-ranks; the
    sharded leg's gather is then staged through the host) and the line says so.  ``force_gloo`` makes the probe fail
    (tests)."""
    import torch.distributed as dist
    control = backend
    if backend == "nccl":
        try:
            if force_gloo:
                raise RuntimeError("RCCL probe failure forced by MCTQ_BENCH_FORCE_GLOO")
            dist.init_process_group("nccl", device_id=device)
            probe = torch.zeros(1, device=device)
            dist.all_reduce(probe)
            torch.cuda.synchronize(device)
        except Exception as e:  # noqa: BLE001
            if not allow_gloo:
                print(f"[bench] RCCL unavailable ({e!r:.300}) and --allow-gloo not given: no multi-GPU line without an RCCL "
                      f"group; exiting with status {EXIT_NO_RCCL}", file=sys.stderr, flush=True)
                try:
                    dist.destroy_process_group()
                except Exception:  # noqa: BLE001
                    pass
                sys.exit(EXIT_NO_RCCL)
            print(f"[bench] RCCL unavailable ({e!r:.200}); --allow-gloo: barrier / max-over-ranks go over gloo", file=sys.stderr, flush=True)
            try:
                dist.destroy_process_group()
            except Exception:  # noqa: BLE001
                pass
            dist.init_process_group("gloo")
            control = "gloo"
    else:
        dist.init_process_group(backend)
    return dist, control


SPIN_FIRST = False      # set by calibrate_sync(): poll the region's last event before torch.cuda.synchronize()?


def sync(device: torch.device, spin_on=None):
    """torch.cuda.synchronize() -- optionally reached by polling an event first.  Which is cheaper depends on the machine:
    where the runtime's blocking wait wakes up late, polling the event recorded behind the last launch saves tens of
    microseconds (round 2's box); where synchronize() on an already idle device costs ~20 us by itself, the plain call is
    cheaper (round 3's box: 458 vs 490 us for a 20-step region, tools/region_overhead_probe.py).  calibrate_sync() times
    both on short regions outside the timed region; the choice is reported in the bench line."""
    if device.type == "cuda":
        if SPIN_FIRST and spin_on is not None:
            while not spin_on.query():
                pass
        torch.cuda.synchronize(device)


def calibrate_sync(step: Callable[[int], None], device: torch.device, steps: int = 10, rounds: int = 5) -> str:
    """Pick the cheaper way to reach torch.cuda.synchronize() at the end of a region (see sync); outside the timed region."""
    global SPIN_FIRST
    if device.type != "cuda":
        return "n/a"
    best = {}
    for spin in (False, True):
        SPIN_FIRST = spin
        ts = []
        for _ in range(rounds):
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            for i in range(steps):
                step(i)
            ev = torch.cuda.Event()
            ev.record()
            sync(device, ev)
            ts.append(time.perf_counter() - t0)
        best[spin] = sorted(ts)[len(ts) // 2]
    SPIN_FIRST = best[True] < best[False]
    return "poll the last event, then torch.cuda.synchronize()" if SPIN_FIRST else "torch.cuda.synchronize()"


def timed_region(step: Callable[[int], None], steps: int, device: torch.device, dist=None) -> float:
    """Wall time of exactly ``steps`` calls of ``step``, bracketed by barrier + device synchronize on both sides.
    The clock stops when THIS rank's device has finished its steps; the closing barrier follows (every rank leaves the
    region together) and the caller takes the MAX over ranks -- the slowest rank's time for its K steps is the job's
    time, and the collective that implements the barrier (tens of microseconds of RCCL, as much as several steps of a
    20-step region) is measurement plumbing, not part of the path."""
    if dist is not None:
        dist.barrier()
    sync(device)
    t0 = time.perf_counter()
    last = None
    for i in range(steps):
        last = step(i)
    sync(device, last if device.type == "cuda" and isinstance(last, torch.cuda.Event) else None)
    elapsed = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
    return elapsed


def max_over_ranks(values, dist, control: str, device: torch.device):
    """Element-wise MAX of a list of floats over all ranks (the slowest rank defines the job's time)."""
    if dist is None:
        return list(values)
    t = torch.tensor(list(values), dtype=torch.float64, device=device if control == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return [float(v) for v in t]


def sharded_cfg5_leg(dist, rank: int, world: int, device: torch.device, rows: int = 8192, cols: int = 8192,
                     reps: int = 60, gather_reps: int = 20, control: Optional[str] = None) -> Dict:
    """BASELINE config 5: WeightsPOT 4-bit per-channel on rows x cols, sharded by dim 0 across the ranks
    (strong scaling: rank r quantizes rows [rows r / N, rows (r+1) / N)), then ONE all-gather (RCCL over xGMI on
    GPUs).  Returns compute-only and compute + all-gather rates; every rank derives the same thresholds from the
    same portable input, so no object collective is needed."""
    from mct_quantizers_amd import workloads
    from mct_quantizers_amd.sharded import ShardedWeightsQuantizer, row_block
    start, stop = row_block(rows, world, rank)
    x_full = workloads.make_input("cfg5", shape=(rows, cols))
    kw = workloads.make_workload("cfg5", x_full).kwargs
    x_local = torch.from_numpy(x_full[start:stop].copy()).to(device)
    del x_full
    sq = ShardedWeightsQuantizer("WeightsPOTInferableQuantizer", kw, full_rows=rows)
    xs = [x_local, x_local.clone(), x_local.clone()]
    y = None
    for i in range(min(6, reps)):
        y = sq(xs[i % 3])
    t_comp = timed_region(lambda i: sq(xs[i % 3]), reps, device, dist) / reps
    y = sq(xs[0])
    # this rank's shard against the REFERENCE's digest of the same row block (tests/golden/shard_sha.json)
    shard_ok = shard_digest_check(y, rows, cols, world, rank)
    staged = device.type == "cuda" and (control or dist.get_backend()) == "gloo"
    if staged:
        # rehearsal on fewer GPUs than ranks (--allow-gloo): gloo has no all-gather of device tensors, so the collective
        # moves host copies of the shards; what is rehearsed is the entry path, the row blocks and the digests, not xGMI
        y = y.cpu()
    # the 256 MiB result lives outside the timed gather.  The collective is issued at world size 1 too (a group of one rank:
    # RCCL copies y into the result), so that all_gather_into_tensor has run on HIP tensors before any multi-GPU run
    buffers = sq.gather_buffers(y)
    full = None
    for _ in range(min(3, gather_reps)):
        full = sq.all_gather(y, buffers, force_collective=True)
    box = {}

    def gather(i):
        box["full"] = sq.all_gather(y, buffers, force_collective=True)
    t_gather = timed_region(gather, gather_reps, device, dist) / gather_reps
    full = box.get("full", full)
    recv = (rows - (stop - start)) * cols * 4
    ok = (full is not None and tuple(full.shape) == (rows, cols) and full.data_ptr() != y.data_ptr()
          and bool(torch.equal(full[start:stop], y)))
    recv_gbs = recv / t_gather / 1e9 if t_gather > 0 else None
    digest = full_digest_check(full, rows, cols) if rank == 0 else None
    shards_ok = gather_flags(dist, shard_ok, control or dist.get_backend(), device)
    return {
        "workload": f"cfg5 WeightsPOT per-channel(axis0) 4b {rows}x{cols}, dim-0 shards",
        "scaling": "strong", "rows_per_rank": stop - start, "ranks": world,
        "compute_ms": t_comp * 1e3, "compute_elems_per_s": rows * cols / t_comp,
        "allgather_ms": t_gather * 1e3, "allgather_recv_bytes_per_rank": recv,
        "allgather_recv_gbs_per_rank": recv_gbs,
        "allgather_backend": dist.get_backend() if dist is not None else None,
        "allgather_device": str(full.device) if full is not None else None,
        "allgather_staged_through_host": staged,
        # a group of one rank moves nothing between GPUs: the figure is RCCL's local copy of the 256 MiB result
        "allgather_local_copy_gbs": (stop - start) * cols * 4 / t_gather / 1e9 if world == 1 and t_gather > 0 else None,
        # xGMI is point to point: a rank receives from its world - 1 peers over world - 1 links at once
        "allgather_gbs_per_link": recv_gbs / (world - 1) if recv_gbs is not None and world > 1 else None,
        "allgather_output": "pre-allocated outside the timed region",
        "compute_plus_allgather_elems_per_s": rows * cols / (t_comp + t_gather),
        "gathered_rows_match_local": ok,
        "gathered_equals_reference_digest": digest,
        # every rank's own shard == the reference's output rows of that block (None: no digest recorded for this shape)
        "ranks_shard_digest_ok": shards_ok,
    }


def sha256_of(t: torch.Tensor) -> str:
    """SHA-256 of a tensor's bytes in its dense order (16-bit types as their bit patterns)."""
    import hashlib
    t = t.detach().cpu().contiguous()
    if t.dtype in (torch.bfloat16, torch.float16):
        t = t.view(torch.int16)
    return hashlib.sha256(t.numpy().tobytes()).hexdigest()


def _golden(name: str) -> dict:
    import json
    # MCTQ_BENCH_GOLDEN_DIR: another directory of digests (tests doctor one to see a mismatch fail the job)
    root = os.environ.get("MCTQ_BENCH_GOLDEN_DIR") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden")
    path = os.path.join(root, name)
    try:
        with open(path) as f:
            return json.load(f)
    except OSError:
        return {}


def golden_output_digest(config: str, shape, batch: int = 8) -> Optional[str]:
    """The reference's SHA-256 of the float32 output of a BASELINE configuration on workloads.py's input
    (tests/golden/full_sha.json, written by tools/gen_golden.py from the imported reference); None if not recorded."""
    configs = _golden("full_sha.json").get("configs", {})
    for key in ((f"cfg3_n{batch}", "cfg3") if config == "cfg3" else (config,)):
        entry = configs.get(key)
        if entry and list(entry.get("shape", [])) == list(shape):
            return entry.get("y_sha256")
    return None


def shard_digest(rows: int, cols: int, world: int, rank: int) -> Optional[str]:
    """The reference's SHA-256 of rank ``rank``'s row block of config 5 at world size ``world``
    (tests/golden/shard_sha.json: the reference's FULL output cut into sharded.row_block's blocks)."""
    rec = _golden("shard_sha.json").get("cfg5", {}).get(f"{rows}x{cols}", {}).get("shards", {}).get(str(world))
    return rec[rank] if rec and rank < len(rec) else None


def shard_digest_check(y_local: torch.Tensor, rows: int, cols: int, world: int, rank: int) -> Optional[bool]:
    want = shard_digest(rows, cols, world, rank)
    return None if want is None else sha256_of(y_local) == want


def gather_flags(dist, flag: Optional[bool], control: Optional[str], device: torch.device):
    """[flag of rank 0, flag of rank 1, ...] on every rank; a flag is True / False / None (nothing to compare with)."""
    code = {True: 1, False: 0, None: -1}[flag]
    if dist is None:
        return [flag]
    t = torch.tensor([code], dtype=torch.int32, device=device if control == "nccl" else "cpu")
    got = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(got, t)
    return [{1: True, 0: False, -1: None}[int(v[0])] for v in got]


def full_digest_check(full: torch.Tensor, rows: int, cols: int) -> Optional[bool]:
    """SHA-256 of the re-assembled tensor against the REFERENCE's digest of BASELINE config 5 (tests/golden/full_sha.json,
    written by tools/gen_golden.py from the reference's output on the same portable input); None when the size has no
    recorded digest (the dry run's small shape)."""
    want = golden_output_digest("cfg5", [rows, cols])
    if want is None:
        want = _golden("shard_sha.json").get("cfg5", {}).get(f"{rows}x{cols}", {}).get("y_sha256")
    if want is None:
        return None
    return sha256_of(full) == want


def traffic_fields(pmc_path: str, key: str, kernel_variant: str, build_id: Optional[str]) -> Dict:
    """The ``roofline.traffic*`` fields of a bench line: measured HBM bytes per launch from the committed PMC passes
    (profiles/pmc_traffic.json, tools/gpu_pmc_traffic.sh + tools/pmc_summarize.py).  The number is emitted only when the
    record was taken on the SAME kernel variant (``mctq_last_launch()``) of the SAME library build (``mctq_build_id()``:
    the content hash of every kernel source, header and compiler flag) as the run that prints the line; otherwise
    ``traffic`` stays null and the line says why (``traffic_build_id_mismatch``: counters of another build are not this
    build's traffic)."""
    import json
    out: Dict = {"traffic": None, "traffic_key": key}
    try:
        with open(pmc_path) as f:
            rec = json.load(f).get(key)
    except OSError:
        out["traffic_source"] = "profiles/pmc_traffic.json missing"
        return out
    if rec is None:
        out["traffic_source"] = f"no PMC record {key!r} in profiles/pmc_traffic.json"
    elif rec.get("variant") != kernel_variant:
        out["traffic_source"] = (f"stale: profiles/pmc_traffic.json was taken on {rec.get('variant')!r}, "
                                 f"this run launched {kernel_variant!r}")
    elif not build_id or rec.get("build_id") != build_id:
        out["traffic_build_id_mismatch"] = True
        out["traffic_recorded_for_another_build"] = rec.get("hbm_bytes_per_launch")
        out["traffic_source"] = (f"profiles/pmc_traffic.json holds counters of library build {rec.get('build_id', 'unrecorded')!r} "
                                 f"(git {rec.get('git_head', '?')}), this run loaded build {build_id!r}: not emitted")
    else:
        out["traffic"] = rec["hbm_bytes_per_launch"]
        out["traffic_build_id_mismatch"] = False
        out["traffic_source"] = (f"profiles/pmc_traffic.json: {rec['variant']} of library build {build_id} at git "
                                 f"{rec.get('git_head', '?')} (FETCH_SIZE x2 + WRITE_SIZE, separate passes)")
    return out


class Watchdog:
    """A stuck collective must not cost the main result -- but it must not look like success either: prints the
    line it was given and exits with status 4 (``status``: the N = 1 line's optional leg on a one-rank group passes 0 --
    there the leg is an extra, and the line without it is complete)."""

    def __init__(self, seconds: float, rank: int, line_fn: Callable[[], str], status: int = 4):
        import threading
        self._t = threading.Timer(seconds, self._fire)
        self._t.daemon = True
        self.rank, self.line_fn, self.status = rank, line_fn, status

    def _fire(self):
        if self.rank == 0:
            print(self.line_fn(), flush=True)
        print(f"[bench] watchdog: the sharded config-5 extras did not finish; exiting with status {self.status}", file=sys.stderr, flush=True)
        os._exit(self.status)

    def __enter__(self):
        self._t.start()
        return self

    def __exit__(self, *exc):
        self._t.cancel()
        return False
