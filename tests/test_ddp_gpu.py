"""Data-parallel training step on real kernels: two processes share the one GPU of the test box and exchange
gradients through `gloo` (RCCL refuses two ranks on one device; the GradReducer code path -- per-layer slices
reported by the backward schedule, bucketed asynchronous all-reduce overlapped with the rest of backward,
finish -- is the one bench.py runs over RCCL).  Checked: both ranks end with identical parameters, and they equal
a single process that saw the two ranks' batches as one (mean of the per-rank mean gradients = global mean when the
ranks count the same number of tokens)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _cfg(batch, variant="plain"):
    from commu_amd.model.config_helper import get_cfg
    if variant == "padded_chunk2":
        # d_model 100 / d_head 50 (zero-padded kernel shapes, the released checkpoint's kind), two micro-batches: the
        # gradient exchange must start only on the LAST micro-batch's backward
        return get_cfg(num_layers=3, num_heads=2, units=100, inner_size=200, tgt_length=32, mem_length=0,
                       batch_size=batch, batch_chunk=2, dropout=0.0, attention_dropout=0.0)
    return get_cfg(num_layers=3, num_heads=2, units=64, inner_size=128, tgt_length=32, mem_length=0, batch_size=batch,
                   batch_chunk=1, dropout=0.0, attention_dropout=0.0)


def _batches(dev, nsteps=3):
    from commu_amd.model.dataset import synthetic_batch
    return [[synthetic_batch(32, 4, dev, seed=300 + 10 * step + r) for r in range(2)] for step in range(nsteps)]


def _worker(rank, world, port, q, variant="plain"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from commu_amd.ddp import GradReducer
        from commu_amd.model.dataset import BaseVocab
        from commu_amd.train import Trainer, build_model
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        cfg = _cfg(8, variant)
        model = build_model(cfg, BaseVocab(), dev, seed=5 + rank)          # different init per rank on purpose
        # several buckets on this small model; "bf16_wire": the gradient crosses the ranks as bf16, summed in fp32
        red = GradReducer(bucket_mb=0.05, wire_dtype="bf16" if variant == "bf16_wire" else "fp32")
        red.broadcast_params(model)
        tr = Trainer(model, cfg, num_gpus=world, reducer=red, graph=(variant == "graph"))
        fired, begins = [], []
        orig_begin = red.begin
        red.begin = lambda: (begins.append(1), orig_begin())[1]
        for step in _batches(dev, 6 if variant == "graph" else 3):
            tr.step(*step[rank])
            fired.append(len(getattr(red, "_fired", [])))
        torch.cuda.synchronize()
        if variant == "graph":
            assert tr.graph_failed is None and tr._graphs is not None, tr.graph_failed
        nll, gnorm, tokens = tr.log_window()              # packed 3-scalar all-reduce (train.py:172-174)
        q.put((rank, {n: p.detach().cpu().numpy() for n, p in model.named_parameters()}, fired,
               (len(begins), nll, gnorm, tokens)))
    except Exception:                                      # surface the worker's traceback in the parent
        import traceback
        q.put((rank, traceback.format_exc(), None, None))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("variant", ["plain", "padded_chunk2", "graph", "bf16_wire"])
def test_two_rank_training_matches_single_process_with_joint_batch(variant):
    """"graph": Trainer(graph=True) on both ranks -- the step replayed from two hipGraphs with the (un-overlapped) gradient
    exchange between them, the path bench.py takes for small per-GPU batches (--global-batch 64 on 8 GPUs)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, variant)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(2)], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
    for r in res:
        assert not isinstance(r[1], str), r[1]
    (_, p0, fired0, log0), (_, p1, _, log1) = res
    nsteps = 6 if variant == "graph" else 3
    if variant == "graph":                                   # (eager steps before the capture use the overlapped exchange)
        assert log0[1:] == log1[1:] and 2 <= log0[0] <= nsteps
    else:
        assert log0 == log1 and log0[0] == 3                 # one exchange per optimiser step; same window on both ranks
    assert log0[3] == 2 * nsteps * 32 * 4                    # tokens of both ranks over the steps
    p0 = {k: torch.from_numpy(v) for k, v in p0.items()}
    p1 = {k: torch.from_numpy(v) for k, v in p1.items()}
    if variant != "graph":
        assert all(n > 1 for n in fired0), fired0            # more than one bucket: the overlapped protocol ran
    for n in p0:
        assert torch.equal(p0[n], p1[n]), n                  # same averaged gradients, same update on both ranks

    # single process, the two ranks' columns side by side, lr = lr/num_gpus as train.py:441 prescribes
    from commu_amd.model.dataset import BaseVocab
    from commu_amd.train import Trainer, build_model
    dev = torch.device("cuda", 0)
    cfg = _cfg(8, variant)
    model = build_model(cfg, BaseVocab(), dev, seed=5)       # rank 0's initial parameters (broadcast source)
    tr = Trainer(model, cfg, num_gpus=2, reducer=None)
    for step in _batches(dev, nsteps):
        (d0, t0, r0, n0), (d1, t1, r1, n1) = step
        assert n0 == n1
        if variant == "padded_chunk2":
            # micro-batch i of the joint run = micro-batch i of rank 0 next to micro-batch i of rank 1
            cat = lambda a, b, dim: torch.cat([x for pair in zip(torch.chunk(a, 2, dim), torch.chunk(b, 2, dim)) for x in pair], dim)
            tr.step(cat(d0, d1, 1), cat(t0, t1, 1), cat(r0, r1, 0), n0 + n1)
        else:
            tr.step(torch.cat([d0, d1], 1), torch.cat([t0, t1], 1), torch.cat([r0, r1], 0), n0 + n1)
    nll1, gnorm1, tok1 = tr.log_window()
    assert tok1 == log0[3] and abs(nll1 - log0[1]) < 2e-3     # token-weighted NLL of the window is rank-count invariant
    for n, p in model.named_parameters():
        a, b = p.detach().cpu(), p0[n]
        # (bf16 wire format: every gradient element carries two bf16 roundings, and Adam turns a relative error of a
        #  small gradient into a step of the same relative size: looser)
        assert torch.allclose(a, b, rtol=0, atol=2e-3 if variant == "bf16_wire" else 3e-4), (n, float((a - b).abs().max()))


def _cfg_mid(batch):
    """Large enough that the grouped weight-gradient launches run on the side stream (T*B >= 4096) and lag the main
    stream: an exchange that starts before its producers have finished would read incomplete gradients."""
    from commu_amd.model.config_helper import get_cfg
    return get_cfg(num_layers=4, num_heads=4, units=256, inner_size=512, tgt_length=256, mem_length=0, batch_size=batch,
                   batch_chunk=1, dropout=0.0, attention_dropout=0.0)


def _rccl_single_worker(rank, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        from commu_amd.ddp import GradReducer
        from commu_amd.model.dataset import BaseVocab, synthetic_batch
        from commu_amd.train import Trainer, build_model
        cfg = _cfg_mid(16)
        batches = [synthetic_batch(256, 16, dev, seed=900 + i) for i in range(3)]
        out = {}
        for mode in ("rccl", "none"):
            model = build_model(cfg, BaseVocab(), dev, seed=5)
            red = GradReducer(bucket_mb=0.5, exchange_single=True) if mode == "rccl" else None
            if red is not None:
                assert red._avg and red._active and red.world == 1
            tr = Trainer(model, cfg, num_gpus=1, reducer=red)
            fired = []
            for b in batches:
                tr.step(*b)
                if red is not None:
                    fired.append(len(red._fired))
            torch.cuda.synchronize()
            out[mode] = {n: p.detach().cpu().numpy() for n, p in model.named_parameters()}
            if red is not None:
                # the un-overlapped form of the same exchange on the final gradients: identity at one rank as well
                g = model._ensure_flat()["g"]
                before = g.clone()
                red.reduce_flat(g, list(model._ensure_flat()["offs"]))
                torch.cuda.synchronize()
                out["reduce_flat_identity"] = bool(torch.equal(before, g))
                out["fired"] = fired
                out["comm_stream"] = red._stream is not None
        q.put((0, out))
    except Exception:
        import traceback
        q.put((0, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_rccl_branch_runs_on_one_device_and_is_the_identity_at_world_1():
    """The `nccl` (= RCCL) branch of GradReducer -- ReduceOp.AVG on bucket slices of the flat gradient, launched from
    the communication stream after the three producer events, finish() joining it -- executed for real in a one-rank
    RCCL group on the test GPU: training with the exchange equals training without it (to fp32 summation-order noise), several buckets
    fire per step, and the un-overlapped reduce_flat is the identity too.  (Two ranks on one device are refused by
    RCCL; the multi-rank arithmetic is covered through gloo above, the ordering by the fake-collective test below.)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_single_worker, args=(0, _free_port(), q))
    p.start()
    _, out = q.get(timeout=600)
    p.join(timeout=60)
    assert not isinstance(out, str), out
    assert out["comm_stream"] and out["reduce_flat_identity"]
    assert all(n > 1 for n in out["fired"]), out["fired"]
    # (not bit for bit: the bias / LayerNorm-parameter column sums end in fp32 atomics, whose order differs run to run)
    import numpy as np
    for n in out["none"]:
        a, b = out["rccl"][n], out["none"][n]
        assert np.abs(a - b).max() <= 1e-6 + 1e-4 * np.abs(b).max(), (n, float(np.abs(a - b).max()))


def test_overlapped_exchange_is_ordered_after_its_producers(monkeypatch):
    """Stream / event ordering of the overlapped exchange on one GPU: torch.distributed.all_reduce is replaced by a
    stand-in that HALVES the slice on the stream it is launched from (from the ordering point of view exactly what a
    collective is: a kernel on the communication stream that rewrites the bucket).  If a bucket were exchanged before
    every kernel that adds into it has run (main stream, weight-gradient stream, reduction stream), the late
    contributions would escape the halving.  Expected: every gradient == 0.5 x the gradient of a run without exchange
    (to the summation-order noise of the column-sum atomics)."""
    from commu_amd import ddp
    from commu_amd.model.dataset import BaseVocab, synthetic_batch
    from commu_amd.train import build_model
    dev = torch.device("cuda", 0)
    cfg = _cfg_mid(16)
    d, t, r, n = synthetic_batch(256, 16, dev, seed=77)
    model_offsets = {}

    def grads(red):
        model = build_model(cfg, BaseVocab(), dev, seed=5)
        model.eval()                                        # (no dropout: both runs see the same function)
        model.zero_grad()
        loss, _ = model(d, t, r, None)
        if red is not None:
            red.begin()
            model.grad_ready_hook = red.range_ready
        loss.float().mean().backward()
        model.grad_ready_hook = None
        fl = model._ensure_flat()
        g = fl["g"]
        if red is not None:
            red.finish(g)
        torch.cuda.synchronize()
        for (n, p_), off in zip(model.named_parameters(), fl["offs"]):
            model_offsets[n] = (off, off + p_.numel())
        return g.clone()

    ref = grads(None)
    launches = []

    class _Done:
        def __init__(self, ev):
            self.ev = ev

        def wait(self):
            torch.cuda.current_stream().wait_event(self.ev)

    def fake_all_reduce(tensor, op=None, group=None, async_op=False):
        launches.append(torch.cuda.current_stream().cuda_stream)
        tensor.mul_(0.5)
        ev = torch.cuda.Event()
        ev.record()
        return _Done(ev)

    monkeypatch.setattr(ddp.dist, "all_reduce", fake_all_reduce)
    red = ddp.GradReducer(bucket_mb=0.5)
    red.world, red._avg, red._active = 2, True, True
    for _ in range(3):
        got = grads(red)
        assert len(red._fired) > 2 and red._stream is not None
        assert set(launches) == {red._stream.cuda_stream}      # every collective was launched from the communication stream
        # (to the summation-order noise of the atomics in the column-sum kernels; an escaped contribution is O(1) of it)
        for name, off in model_offsets.items():
            a, b = got[off[0]:off[1]], ref[off[0]:off[1]] * 0.5
            assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max()) + 1e-12, name


def test_hook_mode_bias_gradients_equal_the_single_stream_pass():
    """Under an overlapped exchange (grad_ready_hook set: no column-sum group) the last layer's bias chain runs on the
    second side stream; the colsum(dq) term of d r_r_bias must not reach the same vector from the first side stream's
    weight-gradient reduction at the same time (round-4 advisor finding).  Repeated hook-mode passes must all equal the
    pass with every kernel on ONE stream."""
    from commu_amd.model.dataset import BaseVocab, synthetic_batch
    from commu_amd.train import build_model
    dev = torch.device("cuda", 0)
    cfg = _cfg_mid(16)
    d, t, r, n = synthetic_batch(256, 16, dev, seed=78)

    def grads(hook, side):
        model = build_model(cfg, BaseVocab(), dev, seed=6)
        model.eval()
        model.wgrad_side_stream = side
        model.zero_grad()
        loss, _ = model(d, t, r, None)
        model.grad_ready_hook = hook
        loss.float().mean().backward()
        model.grad_ready_hook = None
        torch.cuda.synchronize()
        return {k: p.grad.detach().clone() for k, p in model.named_parameters() if k in ("r_r_bias", "r_w_bias")}

    def hook(G, lo, hi, evs=None):          # (orders itself after events, like GradReducer.range_ready: no blocking join)
        return None
    hook.wants_events = True
    ref = grads(None, False)
    for _ in range(5):
        got = grads(hook, True)
        for k in ref:
            a, b = got[k], ref[k]
            assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max()) + 1e-12, k
