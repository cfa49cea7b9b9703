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


def _cfg(batch):
    from commu_amd.model.config_helper import get_cfg
    return get_cfg(num_layers=3, num_heads=2, units=64, inner_size=128, tgt_length=32, mem_length=0, batch_size=batch,
                   batch_chunk=1, dropout=0.0, attention_dropout=0.0)


def _batches(dev):
    from commu_amd.model.dataset import synthetic_batch
    return [[synthetic_batch(32, 4, dev, seed=300 + 10 * step + r) for r in range(2)] for step in range(3)]


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from commu_amd.ddp import GradReducer
        from commu_amd.model.dataset import BaseVocab
        from commu_amd.train import Trainer, build_model
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        cfg = _cfg(8)
        model = build_model(cfg, BaseVocab(), dev, seed=5 + rank)          # different init per rank on purpose
        red = GradReducer(bucket_mb=0.05)                                 # several buckets on this small model
        red.broadcast_params(model)
        tr = Trainer(model, cfg, num_gpus=world, reducer=red)
        fired = []
        for step in _batches(dev):
            tr.step(*step[rank])
            fired.append(len(red._fired))
        torch.cuda.synchronize()
        q.put((rank, {n: p.detach().cpu().numpy() for n, p in model.named_parameters()}, fired))
    except Exception:                                      # surface the worker's traceback in the parent
        import traceback
        q.put((rank, traceback.format_exc(), None))
    finally:
        dist.destroy_process_group()


def test_two_rank_training_matches_single_process_with_joint_batch():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(2)], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
    for r in res:
        assert not isinstance(r[1], str), r[1]
    (_, p0, fired0), (_, p1, _) = res
    p0 = {k: torch.from_numpy(v) for k, v in p0.items()}
    p1 = {k: torch.from_numpy(v) for k, v in p1.items()}
    assert all(n > 1 for n in fired0), fired0                # more than one bucket: the overlapped protocol ran
    for n in p0:
        assert torch.equal(p0[n], p1[n]), n                  # same averaged gradients, same update on both ranks

    # single process, the two ranks' columns side by side, lr = lr/num_gpus as train.py:441 prescribes
    from commu_amd.model.dataset import BaseVocab
    from commu_amd.train import Trainer, build_model
    dev = torch.device("cuda", 0)
    cfg = _cfg(8)
    model = build_model(cfg, BaseVocab(), dev, seed=5)       # rank 0's initial parameters (broadcast source)
    tr = Trainer(model, cfg, num_gpus=2, reducer=None)
    for step in _batches(dev):
        (d0, t0, r0, n0), (d1, t1, r1, n1) = step
        assert n0 == n1
        tr.step(torch.cat([d0, d1], 1), torch.cat([t0, t1], 1), torch.cat([r0, r1], 0), n0 + n1)
    for n, p in model.named_parameters():
        a, b = p.detach().cpu(), p0[n]
        assert torch.allclose(a, b, rtol=0, atol=3e-4), (n, float((a - b).abs().max()))
