"""Dim-0 sharding of a large weight across the GPUs of one node (one process per GPU).

The quantizers are elementwise given the per-channel parameters, so a weight splits into
contiguous row blocks with NO exchange during compute.  With ``channel_axis == 0`` the parameter
vectors split the same way; with any other axis they are replicated.  Re-assembly, when a consumer
needs the whole tensor on every rank, is ONE ``all_gather_into_tensor`` (RCCL over xGMI through
``torch.distributed``, backend "nccl"; "gloo" on CPU for tests).  The reference has nothing like
this: it is new, and optional.

    sq = ShardedWeightsQuantizer("WeightsPOTInferableQuantizer",
                                 dict(num_bits=4, threshold=thr, per_channel=True, channel_axis=0),
                                 full_rows=8192)
    y_local = sq(w_local)            # rows [sq.start, sq.stop) of the full weight
    y_full = sq.all_gather(y_local)  # optional
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.distributed as dist

from mct_quantizers_amd.pytorch import quantizers as _q

_PER_CHANNEL_LISTS = ("threshold", "min_range", "max_range")


def row_block(full_rows: int, world: int, rank: int) -> Tuple[int, int]:
    """Rows [start, stop) owned by ``rank``: equal blocks of ceil(full_rows / world), last ones may be short."""
    per = -(-full_rows // world)
    start = min(full_rows, rank * per)
    return start, min(full_rows, start + per)


def shard_kwargs(kwargs: dict, full_rows: int, world: int, rank: int) -> dict:
    """Constructor kwargs of the rank-local quantizer: per-channel lists are sliced when channel_axis == 0."""
    out = dict(kwargs)
    if out.get("per_channel") and out.get("channel_axis") == 0:
        start, stop = row_block(full_rows, world, rank)
        for key in _PER_CHANNEL_LISTS:
            if key in out and len(out[key]) == full_rows:
                out[key] = list(out[key][start:stop])
    return out


class ShardedWeightsQuantizer:
    """Rank-local view of a weights quantizer applied to a dim-0 sharded tensor."""

    def __init__(self, quantizer: str, kwargs: dict, full_rows: int, group: Optional[dist.ProcessGroup] = None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.full_rows = full_rows
        self.start, self.stop = row_block(full_rows, self.world, self.rank)
        self.rows_per_rank = -(-full_rows // self.world)
        if self.stop > self.start:
            self.quantizer = getattr(_q, quantizer)(**shard_kwargs(kwargs, full_rows, self.world, self.rank))
        else:
            self.quantizer = None                       # more ranks than rows: this rank owns nothing

    def local_rows(self) -> Tuple[int, int]:
        return self.start, self.stop

    def __call__(self, w_local: torch.Tensor) -> torch.Tensor:
        if w_local.shape[0] != self.stop - self.start:
            raise ValueError(f"rank {self.rank} owns rows [{self.start}, {self.stop}) but got {w_local.shape[0]} rows")
        if self.quantizer is None:
            return torch.empty_like(w_local)
        return self.quantizer(w_local)

    def gather_buffers(self, y_local: torch.Tensor):
        """(out, pad): the full-size result tensor and, when this rank's block is shorter than the common block size,
        a staging tensor of the common size -- allocate once, pass to ``all_gather`` every time."""
        tail = tuple(y_local.shape[1:])
        out = torch.empty((self.rows_per_rank * self.world,) + tail, dtype=y_local.dtype, device=y_local.device)
        pad = None
        if y_local.shape[0] != self.rows_per_rank:
            pad = torch.zeros((self.rows_per_rank,) + tail, dtype=y_local.dtype, device=y_local.device)
        return out, pad

    def all_gather(self, y_local: torch.Tensor, buffers=None, force_collective: bool = False) -> torch.Tensor:
        """Concatenate every rank's row block along dim 0 (one collective; every rank gets the full tensor).
        ``buffers``: the pair from ``gather_buffers`` (no allocation inside the call); default: allocate.
        ``force_collective``: issue ``all_gather_into_tensor`` even in a group of ONE rank (the result is then a copy
        of ``y_local``) -- the benchmark's way of running RCCL on device memory on a one-GPU machine."""
        if self.world == 1 and not (force_collective and dist.is_initialized()):
            return y_local
        out, pad = buffers if buffers is not None else self.gather_buffers(y_local)
        if y_local.shape[0] != self.rows_per_rank:      # short last block: pad to the common size
            pad[: y_local.shape[0]] = y_local
            y_local = pad
        dist.all_gather_into_tensor(out, y_local.contiguous(), group=self.group)
        return out[: self.full_rows]
