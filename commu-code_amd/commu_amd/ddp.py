"""Data-parallel gradient exchange for the training step (reference: DistributedDataParallel at
train.py:467-473 and the scalar all-reduces of :172-174,209-210).

One process per GPU; the gradients live in ONE flat fp32 buffer, so the exchange is a handful of
large all-reduces over RCCL/xGMI (`backend="nccl"` is RCCL on ROCm) issued once per optimiser
step, bucket by bucket in the order the backward pass finishes them, on a side stream.  The
same code runs on CPU tensors with the `gloo` backend (tests).
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def make_buckets(total: int, boundaries: List[int], target_elems: int) -> List[Tuple[int, int]]:
    """Split [0, total) at the given parameter boundaries into contiguous ranges of about
    `target_elems` elements (never splitting inside a boundary interval unless it is larger)."""
    cuts = sorted(set([0, total] + [b for b in boundaries if 0 < b < total]))
    out, start = [], 0
    for i in range(1, len(cuts)):
        if cuts[i] - start >= target_elems or i == len(cuts) - 1:
            out.append((start, cuts[i]))
            start = cuts[i]
    return [(a, b) for a, b in out if b > a]


class GradReducer:
    """Mean all-reduce of a flat gradient buffer in buckets."""

    def __init__(self, group=None, bucket_mb: float = 16.0):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.bucket_elems = int(bucket_mb * (1 << 20) / 4)
        self._stream = None

    def reduce_flat(self, flat_g: torch.Tensor, boundaries: Optional[List[int]] = None):
        if self.world == 1:
            return
        buckets = make_buckets(flat_g.numel(), boundaries or [], self.bucket_elems)
        avg = flat_g.is_cuda            # RCCL reduces to the mean directly; gloo has no AVG
        op = dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM
        handles = []
        # reverse order: the backward pass finishes the LAST parameters first
        for a, b in reversed(buckets):
            handles.append(dist.all_reduce(flat_g[a:b], op=op, group=self.group, async_op=True))
        for h in handles:
            h.wait()
        if not avg:
            flat_g.mul_(1.0 / self.world)

    def allreduce_mean(self, model):
        fl = model._ensure_flat()
        self.reduce_flat(fl["g"], list(fl["offs"]))

    def broadcast_params(self, model, src=0):
        """DDP constructor semantics: rank 0's parameters everywhere (train.py:467)."""
        if self.world == 1:
            return
        fl = model._ensure_flat()
        dist.broadcast(fl["p"], src=src, group=self.group)
        model._refresh_shadows()
