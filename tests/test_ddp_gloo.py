"""Data-parallel gradient exchange on CPU: two processes, `gloo` backend (the same GradReducer
code path runs over RCCL on the GPUs).  Checks the bucketed mean all-reduce of the flat gradient
buffer, the parameter broadcast semantics and the reference's per-rank conventions
(lr / num_gpus, seed + 1000 * rank: train.py:394,441)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from commu_amd.ddp import GradReducer, make_buckets


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n = 100_003
        offs = list(range(0, n, 9973))
        g = torch.arange(n, dtype=torch.float32) * (rank + 1)
        red = GradReducer(bucket_mb=0.05)
        red.reduce_flat(g, offs)
        expect = torch.arange(n, dtype=torch.float32) * (sum(range(1, world + 1)) / world)
        ok = torch.allclose(g, expect, rtol=1e-6)

        class FakeModel:                       # the reducer only needs the flat buffers
            def __init__(self):
                self.fl = {"p": torch.full((1000,), float(rank)), "g": torch.full((1000,), float(rank + 1)),
                           "offs": [0, 500], "params": []}
                self.refreshed = 0

            def _ensure_flat(self):
                return self.fl

            def _refresh_shadows(self):
                self.refreshed += 1
        m = FakeModel()
        red.broadcast_params(m, src=0)
        ok = ok and bool((m.fl["p"] == 0).all()) and m.refreshed == 1
        red.allreduce_mean(m)
        ok = ok and torch.allclose(m.fl["g"], torch.full((1000,), (1 + world) / 2.0))
        # overlapped protocol: layer slices arrive top-down while "backward" is still running; head and tail
        # (shared biases + embedding, output bias) are only covered by finish()
        n2 = 50_000
        g2 = torch.arange(n2, dtype=torch.float32) * (rank + 1)
        red2 = GradReducer(bucket_mb=0.05)          # 13107 elements per bucket
        red2.begin()
        for lo, hi in [(40_000, 49_000), (31_000, 40_000), (22_000, 31_000), (13_000, 22_000), (4_000, 13_000)]:
            red2.range_ready(g2, lo, hi)
        fired_early = list(red2._fired)
        red2.finish(g2)
        expect2 = torch.arange(n2, dtype=torch.float32) * (sum(range(1, world + 1)) / world)
        ok = ok and torch.allclose(g2, expect2, rtol=1e-6)
        ok = ok and fired_early == [(31_000, 49_000), (13_000, 31_000)]      # two merged buckets before finish
        # collective yes/no (graph capture: one rank's failure takes every rank to the eager step)
        ok = ok and red.all_ok(True) and not red.all_ok(rank != 1) and not red.all_ok(False)
        cover = sorted(red2._fired)
        ok = ok and cover[0][0] == 0 and cover[-1][1] == n2 and all(a[1] == b[0] for a, b in zip(cover, cover[1:]))
        # bf16 WIRE FORMAT, fp32 accumulation (GradReducer(wire_dtype="bf16")): all-to-all of bf16 shards, fp32 sum on
        # the owning rank, all-gather of the rounded means.  Expected value per element: bf16(mean of the bf16-rounded
        # addends, summed in fp32) -- computed here the same way; both ranks must end bit-identical.
        gen = torch.Generator().manual_seed(5)
        base = [torch.randn(n2, generator=torch.Generator().manual_seed(100 + r)) * 0.01 for r in range(world)]
        g3 = base[rank].clone()
        red3 = GradReducer(bucket_mb=0.05, wire_dtype="bf16")
        red3.begin()
        for lo, hi in [(40_000, 49_000), (31_000, 40_000), (22_000, 31_000), (13_000, 22_000), (4_000, 13_000)]:
            red3.range_ready(g3, lo, hi)
        red3.finish(g3)
        want = (sum(b.to(torch.bfloat16).float() for b in base) * (1.0 / world)).to(torch.bfloat16).float()
        ok = ok and torch.equal(g3, want)
        ok = ok and red3.wire_bytes_step == sum(int(2 * (world - 1) / world * (b - a) * 2) for a, b in red3._fired)
        g4 = base[rank].clone()
        red3.reduce_flat(g4, list(range(0, n2, 7001)))
        ok = ok and torch.equal(g4, want)
        ok = ok and red.wire_bytes_step > 0 and red3.comm_stats()["exposed_ms_per_step"] is None      # (CPU tensors: no events)
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


def test_bucketed_mean_allreduce_two_ranks():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]


def test_make_buckets_cover_everything_once():
    total = 1000
    bounds = [0, 100, 250, 260, 700, 990]
    for target in (1, 50, 300, 5000):
        b = make_buckets(total, bounds, target)
        assert b[0][0] == 0 and b[-1][1] == total
        for (a0, a1), (b0, b1) in zip(b, b[1:]):
            assert a1 == b0 and a1 > a0
        assert all(lo in bounds + [total] for lo, _ in b)


def test_single_process_is_a_noop():
    red = GradReducer()
    g = torch.ones(10)
    red.reduce_flat(g, [0, 5])
    assert bool((g == 1).all())
