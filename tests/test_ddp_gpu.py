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


def _batches(dev):
    from commu_amd.model.dataset import synthetic_batch
    return [[synthetic_batch(32, 4, dev, seed=300 + 10 * step + r) for r in range(2)] for step in range(3)]


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
        red = GradReducer(bucket_mb=0.05)                                 # several buckets on this small model
        red.broadcast_params(model)
        tr = Trainer(model, cfg, num_gpus=world, reducer=red)
        fired, begins = [], []
        orig_begin = red.begin
        red.begin = lambda: (begins.append(1), orig_begin())[1]
        for step in _batches(dev):
            tr.step(*step[rank])
            fired.append(len(red._fired))
        torch.cuda.synchronize()
        nll, gnorm, tokens = tr.log_window()              # packed 3-scalar all-reduce (train.py:172-174)
        q.put((rank, {n: p.detach().cpu().numpy() for n, p in model.named_parameters()}, fired,
               (len(begins), nll, gnorm, tokens)))
    except Exception:                                      # surface the worker's traceback in the parent
        import traceback
        q.put((rank, traceback.format_exc(), None, None))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("variant", ["plain", "padded_chunk2"])
def test_two_rank_training_matches_single_process_with_joint_batch(variant):
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
    assert log0 == log1 and log0[0] == 3                     # one exchange per optimiser step; same window on both ranks
    assert log0[3] == 2 * 3 * 32 * 4                         # tokens of both ranks over the three steps
    p0 = {k: torch.from_numpy(v) for k, v in p0.items()}
    p1 = {k: torch.from_numpy(v) for k, v in p1.items()}
    assert all(n > 1 for n in fired0), fired0                # more than one bucket: the overlapped protocol ran
    for n in p0:
        assert torch.equal(p0[n], p1[n]), n                  # same averaged gradients, same update on both ranks

    # single process, the two ranks' columns side by side, lr = lr/num_gpus as train.py:441 prescribes
    from commu_amd.model.dataset import BaseVocab
    from commu_amd.train import Trainer, build_model
    dev = torch.device("cuda", 0)
    cfg = _cfg(8, variant)
    model = build_model(cfg, BaseVocab(), dev, seed=5)       # rank 0's initial parameters (broadcast source)
    tr = Trainer(model, cfg, num_gpus=2, reducer=None)
    for step in _batches(dev):
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
        assert torch.allclose(a, b, rtol=0, atol=3e-4), (n, float((a - b).abs().max()))
