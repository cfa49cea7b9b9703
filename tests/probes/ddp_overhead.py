"""What the data-parallel code path costs on ONE GPU: the bench-shape step with the overlapped gradient exchange active in a
one-rank RCCL group (the all-reduces move nothing) against the plain single-process step."""
import os, sys, time, torch
import torch.distributed as dist
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "commu-code_amd"))
from commu_amd.model.config_helper import get_cfg
from commu_amd.model.dataset import BaseVocab, synthetic_batch
from commu_amd.train import Trainer, build_model
from commu_amd.ddp import GradReducer
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
dist.init_process_group(backend="nccl", init_method="env://", rank=0, world_size=1)
B = int(os.environ.get("DDP_B", 64))
cfg = get_cfg(num_layers=6, num_heads=8, units=512, inner_size=1024, tgt_length=1024, mem_length=0, batch_size=B, batch_chunk=1,
              dropout=0.1, attention_dropout=0.1)
for mode in os.environ.get("DDP_MODES", "plain,reducer,single,plain,reducer,single").split(","):
    model = build_model(cfg, BaseVocab(), dev, seed=3)
    model.train()
    # reducer: the hooks of the overlapped exchange without any collective (a one-rank reducer is inactive);
    # single: the whole protocol -- buckets, communication stream, RCCL all-reduces of one rank
    red = None if mode == "plain" else GradReducer(exchange_single=(mode == "single"))
    tr = Trainer(model, cfg, num_gpus=1, reducer=red, graph=False)
    d, t, r, n = synthetic_batch(1024, B, dev, seed=9)
    for _ in range(5):
        tr.step(d, t, r, n)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        tr.step(d, t, r, n)
    torch.cuda.synchronize()
    print(f"{mode:8s} {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per step", flush=True)
    del tr, model
dist.destroy_process_group()
