"""The N>1 path on CPU: two processes, gloo backend, dim-0 shards + all-gather == unsharded result."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, rows, cols, q):
    import logging
    logging.getLogger("mct_quantizers_amd").setLevel(logging.ERROR)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mct_quantizers_amd import workloads
        from mct_quantizers_amd.pytorch import quantizers as Q
        from mct_quantizers_amd.sharded import ShardedWeightsQuantizer, row_block
        out = {}
        for cfg, name in (("cfg5", "WeightsPOTInferableQuantizer"), ("cfg4", "WeightsLUTSymmetricInferableQuantizer"),
                          ("cfg2", "WeightsSymmetricInferableQuantizer")):
            x_np = workloads.make_input(cfg, shape=(rows, cols))
            wl = workloads.make_workload(cfg, x_np)
            full_want = getattr(Q, wl.quantizer)(**wl.kwargs)(torch.from_numpy(x_np))
            sq = ShardedWeightsQuantizer(wl.quantizer, wl.kwargs, full_rows=rows)
            start, stop = sq.local_rows()
            assert (start, stop) == row_block(rows, world, rank)
            y_local = sq(torch.from_numpy(x_np[start:stop].copy()))
            assert torch.equal(y_local, full_want[start:stop])            # no exchange needed for compute
            y_full = sq.all_gather(y_local)
            assert y_full.shape == full_want.shape and torch.equal(y_full, full_want)
            out[cfg] = float(y_full.double().sum())
        # replicated parameters when the channel axis is not the sharded one
        x = torch.from_numpy(workloads.make_input("cfg2", shape=(rows, 6)))
        kw = dict(num_bits=8, threshold=[0.5, 1.0, 1.5, 2.0, 2.5, 3.0], per_channel=True, channel_axis=1)
        want = Q.WeightsSymmetricInferableQuantizer(**kw)(x.clone())
        sq = ShardedWeightsQuantizer("WeightsSymmetricInferableQuantizer", kw, full_rows=rows)
        start, stop = sq.local_rows()
        assert torch.equal(sq.all_gather(sq(x[start:stop].clone())), want)
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("rows", [16, 13])          # even split, and a short last block
def test_two_rank_shards_reassemble_to_the_unsharded_result(rows):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, rows, 64, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=240)
        assert p.exitcode == 0
    results = dict(q.get(timeout=10) for _ in range(2))
    assert results[0] == results[1]


def test_row_blocks_cover_everything():
    from mct_quantizers_amd.sharded import row_block, shard_kwargs
    for rows in (1, 7, 8, 8192, 8193):
        for world in (1, 2, 4, 8):
            blocks = [row_block(rows, world, r) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == rows
            assert all(a[1] == b[0] for a, b in zip(blocks, blocks[1:]))
    kw = dict(num_bits=4, threshold=list(range(1, 9)), per_channel=True, channel_axis=0)
    assert shard_kwargs(kw, 8, 4, 2)["threshold"] == [5, 6]
    kw1 = dict(kw, channel_axis=1)
    assert shard_kwargs(kw1, 8, 4, 2)["threshold"] == list(range(1, 9))
