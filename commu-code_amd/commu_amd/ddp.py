"""Data-parallel gradient exchange for the training step (reference: DistributedDataParallel at
train.py:467-473 and the scalar all-reduces of :172-174,209-210).

One process per GPU; the gradients live in ONE flat fp32 buffer, so the exchange is a handful of
large all-reduces over RCCL/xGMI (`backend="nccl"` is RCCL on ROCm), once per optimiser step.
OVERLAP: the backward schedule finishes the layers top-down and reports each layer's slice of the
flat buffer through `model.grad_ready_hook`; slices are merged into buckets and every full bucket
is all-reduced asynchronously right away (RCCL's stream orders itself after the kernels enqueued
so far and runs beside the rest of the backward pass); `finish` exchanges what is left (embedding,
shared biases, a partial bucket) and waits.  The same code runs on CPU tensors with `gloo` (tests).
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

    def __init__(self, group=None, bucket_mb: float = 16.0, exchange_single: bool = False):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.bucket_elems = int(bucket_mb * (1 << 20) / 4)
        self._stream = None
        # RCCL reduces to the mean directly; gloo (CPU tests, or CUDA tensors staged through the host) has no AVG
        self._avg = dist.is_initialized() and dist.get_backend(group) == "nccl"
        # exchange_single: run the whole exchange protocol (buckets, communication stream, collectives) in a group of
        # ONE rank too -- the mean over one rank is the identity, so this only exists to execute the RCCL path on a
        # single device (tests/test_ddp_gpu.py); a one-rank job otherwise skips the exchange.
        self._active = self.world > 1 or (exchange_single and dist.is_initialized())

    def reduce_flat(self, flat_g: torch.Tensor, boundaries: Optional[List[int]] = None):
        if not self._active:
            return
        buckets = make_buckets(flat_g.numel(), boundaries or [], self.bucket_elems)
        avg = self._avg
        op = dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM
        handles = []
        # reverse order: the backward pass finishes the LAST parameters first
        for a, b in reversed(buckets):
            handles.append(dist.all_reduce(flat_g[a:b], op=op, group=self.group, async_op=True))
        for h in handles:
            h.wait()
        if not avg:
            flat_g.mul_(1.0 / self.world)

    # ---- overlapped exchange (one optimiser step): begin -> range_ready* -> finish
    def begin(self):
        self._handles, self._fired, self._pending, self._events = [], [], None, []

    def _fire(self, flat_g, a, b):
        """Launch the all-reduce of flat_g[a:b], ordered after the pending producer events.  Device tensors: the
        collective is launched from a dedicated COMMUNICATION STREAM that waits for those events -- whatever the
        backend: RCCL orders a collective after the stream it is launched from, and gloo's device-tensor path
        synchronises with that stream before staging through the host -- so neither the main stream nor the
        weight-gradient streams are ever held up by the exchange."""
        op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
        events, self._events = self._events, []
        if flat_g.is_cuda:
            if self._stream is None:
                self._stream = torch.cuda.Stream(device=flat_g.device)
            if not events:                       # no producer events given: everything enqueued on the caller's stream so far
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(flat_g.device))
                events = [ev]
            with torch.cuda.stream(self._stream):
                for ev in events:
                    self._stream.wait_event(ev)
                self._handles.append(dist.all_reduce(flat_g[a:b], op=op, group=self.group, async_op=True))
        else:
            self._handles.append(dist.all_reduce(flat_g[a:b], op=op, group=self.group, async_op=True))
        self._fired.append((a, b))

    def range_ready(self, flat_g, lo, hi, events=None):
        """model.grad_ready_hook: flat_g[lo:hi] is final once `events` (one per producing stream; None: now) have
        completed.  Ranges arrive top layer first (descending)."""
        if not self._active:
            return
        if self._pending is not None and self._pending[0] == hi:
            self._pending = (lo, self._pending[1])
        else:
            if self._pending is not None:
                self._fire(flat_g, *self._pending)
            self._pending = (lo, hi)
        if events:
            self._events.extend(events)
        if self._pending[1] - self._pending[0] >= self.bucket_elems:
            self._fire(flat_g, *self._pending)
            self._pending = None
    range_ready.wants_events = True

    def finish(self, flat_g):
        if not self._active:
            return
        if self._pending is not None:
            self._fire(flat_g, *self._pending)
            self._pending = None
        pos = 0
        for a, b in sorted(self._fired) + [(flat_g.numel(), flat_g.numel())]:      # the uncovered rest
            if a > pos:
                self._fire(flat_g, pos, a)
            pos = max(pos, b)
        for h in self._handles:
            h.wait()                             # (device tensors: the CURRENT stream waits for the collective)
        if flat_g.is_cuda and self._stream is not None:
            torch.cuda.current_stream(flat_g.device).wait_stream(self._stream)
        if not self._avg:
            flat_g.mul_(1.0 / self.world)
        self._handles = []

    # ---- scalar reductions of the training loop (train.py:172-174, 209-210, 264-265): ONE packed all-reduce each
    def sum_scalars(self, values, device=None):
        """Sum of each entry of `values` (python numbers or 0-d tensors) over the ranks, as python floats."""
        t = torch.stack([torch.as_tensor(v, dtype=torch.float64, device=device).reshape(()) for v in values])
        if self.world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return [float(x) for x in t.cpu()]

    def all_ok(self, ok: bool, device=None) -> bool:
        """True iff `ok` holds on EVERY rank (one MIN all-reduce): for decisions all ranks must take together, e.g.
        replayed graphs against eager steps, whose gradient exchanges issue different collectives."""
        if self.world == 1:
            return bool(ok)
        t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
        return bool(int(t.item()))

    def barrier(self):
        if self.world > 1:
            dist.barrier(group=self.group)

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
