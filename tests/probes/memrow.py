"""Per-step (synchronised) times of a bench shape: when does the step time reach its steady state?"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "commu-code_amd"))
import torch
from commu_amd.model.config_helper import get_cfg
from commu_amd.model.dataset import BaseVocab, synthetic_batch
from commu_amd.train import Trainer, build_model
dev = torch.device("cuda", 0)
def run(n=14, **kw):
    c = dict(num_layers=6, num_heads=8, units=512, inner_size=1024, tgt_length=1024, mem_length=0, batch_size=64)
    c.update(kw)
    cfg = get_cfg(batch_chunk=1, dropout=0.1, attention_dropout=0.1, **c)
    model = build_model(cfg, BaseVocab(), dev, seed=1); model.train()
    tr = Trainer(model, cfg, num_gpus=1)
    batches = [synthetic_batch(c["tgt_length"], c["batch_size"], dev, seed=i) for i in range(4)]
    ts = []
    for i in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        tr.step(*batches[i % 4]); torch.cuda.synchronize()
        ts.append(round(1e3 * (time.perf_counter() - t0), 1))
    print(kw, "per-step ms (synced):", ts, " peak GB", round(torch.cuda.max_memory_allocated() / 2**30, 1), flush=True)
    del tr, model, batches; torch.cuda.empty_cache()
run(); run(mem_length=1024)
cfg5 = dict(num_layers=12, num_heads=16, units=1024, inner_size=2048, tgt_length=2048, mem_length=2048, batch_size=8)
run(**cfg5); run(**cfg5)
