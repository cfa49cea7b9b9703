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

    def __init__(self, group=None, bucket_mb: float = 16.0, exchange_single: bool = False, wire_dtype: str = "fp32"):
        """wire_dtype "fp32": one mean all-reduce per bucket on the fp32 gradient itself (58 MB per step at the bench
        model).  "bf16" (SURVEY C4 allows a 16-bit exchange): the bucket crosses the links as bf16 but is SUMMED IN FP32 --
        an all-to-all hands rank j the j-th shard of every rank's bucket (bf16), rank j adds the `world` shards in fp32,
        and an all-gather returns the rounded means: 2 (N-1)/N x 2 bytes per element on the wire, half of the fp32
        all-reduce, with one rounding of each addend and one of the result instead of a bf16 running sum inside the
        collective; every rank ends with bit-identical gradients (they all receive the same shards).  xGMI is a full mesh
        of point-to-point links, which is exactly what an all-to-all uses."""
        if wire_dtype not in ("fp32", "bf16"):
            raise ValueError("wire_dtype must be 'fp32' or 'bf16'")
        self.wire_dtype = wire_dtype
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.bucket_elems = int(bucket_mb * (1 << 20) / 4)
        self._stream = None
        # RCCL reduces to the mean directly; gloo (CPU tests, or CUDA tensors staged through the host) has no AVG
        self.backend = dist.get_backend(group) if dist.is_initialized() else None
        self._avg = self.backend == "nccl"
        # gloo moves host memory only: device tensors (two test ranks on one GPU) are staged through the host
        self._host_staged = self.backend == "gloo"
        self._wire_buf = {}          # bf16 wire format: (elements, device) -> (send, recv, mean, out), allocated once per bucket size
        # exchange_single: run the whole exchange protocol (buckets, communication stream, collectives) in a group of
        # ONE rank too -- the mean over one rank is the identity, so this only exists to execute the RCCL path on a
        # single device (tests/test_ddp_gpu.py); a one-rank job otherwise skips the exchange.
        self._active = self.world > 1 or (exchange_single and dist.is_initialized())
        # attribution for the scaling runs: bytes each rank puts on the wire per step, and how long the main stream sat
        # in finish() waiting for the exchange (event pairs, read back by comm_stats() after a synchronize)
        self.wire_bytes_step = 0
        self._exposed = []

    def _wire_bytes(self, n):
        w = max(self.world, 1)
        if self.wire_dtype == "bf16":          # what is SENT: the zero-padded shards (w * ceil(n / w) elements)
            n = w * ((n + w - 1) // w)
        return int(2 * (w - 1) / w * n * (2 if self.wire_dtype == "bf16" else 4))

    def _exchange(self, t, async_op):
        """Mean of the fp32 slice `t` over the ranks, in place; returns a handle (or None when it ran synchronously on the
        current stream / thread)."""
        self.wire_bytes_step += self._wire_bytes(t.numel())
        if self.wire_dtype == "fp32":
            op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
            return dist.all_reduce(t, op=op, group=self.group, async_op=async_op)
        w = self.world
        n = t.numel()
        shard = (n + w - 1) // w
        staged = t.is_cuda and self._host_staged
        # the staging buffers of a bucket size live across steps (a step used to allocate four tensors per bucket on the
        # communication stream); the padding of the last shard is zeroed once, at allocation
        key = (n, "cpu" if staged else str(t.device))
        bufs = self._wire_buf.get(key)
        if bufs is None:
            dev = "cpu" if staged else t.device
            bufs = (torch.zeros(w * shard, dtype=torch.bfloat16, device=dev), torch.empty(w * shard, dtype=torch.bfloat16, device=dev),
                    torch.empty(shard, dtype=torch.bfloat16, device=dev), torch.empty(w * shard, dtype=torch.bfloat16, device=dev))
            self._wire_buf[key] = bufs
        send, recv, mean, out = bufs
        if staged:
            # stage the bf16 shards through the host, ordered after the stream this runs on (RCCL takes the device
            # tensors as they are)
            torch.cuda.current_stream(t.device).synchronize()
            send[:n].copy_(t.to(torch.bfloat16))
        else:
            send[:n].copy_(t)
        dist.all_to_all_single(recv, send, group=self.group)
        mean.copy_(recv.view(w, shard).float().sum(0) * (1.0 / w))
        dist.all_gather_into_tensor(out, mean, group=self.group)
        t.copy_(out[:n])
        return None

    def reduce_flat(self, flat_g: torch.Tensor, boundaries: Optional[List[int]] = None):
        if not self._active:
            return
        buckets = make_buckets(flat_g.numel(), boundaries or [], self.bucket_elems)
        self.wire_bytes_step = 0
        handles = []
        # reverse order: the backward pass finishes the LAST parameters first
        for a, b in reversed(buckets):
            handles.append(self._exchange(flat_g[a:b], True))
        for h in handles:
            if h is not None:
                h.wait()
        if self.wire_dtype == "fp32" and not self._avg:
            flat_g.mul_(1.0 / self.world)

    # ---- overlapped exchange (one optimiser step): begin -> range_ready* -> finish
    def begin(self):
        self._handles, self._fired, self._pending, self._events = [], [], None, []
        self.wire_bytes_step = 0

    def _fire(self, flat_g, a, b):
        """Launch the all-reduce of flat_g[a:b], ordered after the pending producer events.  Device tensors: the
        collective is launched from a dedicated COMMUNICATION STREAM that waits for those events -- whatever the
        backend: RCCL orders a collective after the stream it is launched from, and gloo's device-tensor path
        synchronises with that stream before staging through the host -- so neither the main stream nor the
        weight-gradient streams are ever held up by the exchange."""
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
                self._handles.append(self._exchange(flat_g[a:b], True))
        else:
            self._handles.append(self._exchange(flat_g[a:b], True))
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
        e0 = None
        if flat_g.is_cuda:
            cur = torch.cuda.current_stream(flat_g.device)
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record(cur)
        for h in self._handles:
            if h is not None:
                h.wait()                         # (device tensors: the CURRENT stream waits for the collective)
        if flat_g.is_cuda and self._stream is not None:
            torch.cuda.current_stream(flat_g.device).wait_stream(self._stream)
        if e0 is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record(cur)
            self._exposed.append((e0, e1))
            if len(self._exposed) > 4096:
                self._exposed = self._exposed[-2048:]
        if self.wire_dtype == "fp32" and not self._avg:
            flat_g.mul_(1.0 / self.world)
        self._handles = []

    def comm_stats(self, last: Optional[int] = None):
        """{"wire_bytes_per_step", "exposed_ms_per_step", "steps"}: bytes this rank sent per optimiser step (ring
        all-reduce or all-to-all + all-gather volume, 2 (N-1)/N x payload) and the mean time the main stream spent in
        finish() between arriving and being released by the last collective -- the part of the exchange the backward
        pass did NOT hide.  Call after a device synchronize; `last`: only the most recent steps."""
        pairs = self._exposed if last is None else self._exposed[-last:]
        ms = [a.elapsed_time(b) for a, b in pairs]
        return {"wire_bytes_per_step": int(self.wire_bytes_step), "wire_dtype": self.wire_dtype,
                "exposed_ms_per_step": (sum(ms) / len(ms)) if ms else None, "steps": len(ms)}

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
