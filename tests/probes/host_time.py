"""Host (Python + launch) time per training step vs the GPU time of the same steps: is the bench host-bound?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "commu-code_amd")); sys.path.insert(0, ROOT)
import torch
from commu_amd.model.config_helper import get_cfg
from commu_amd.model.dataset import BaseVocab, synthetic_batch
from commu_amd.train import Trainer, build_model
dev = torch.device("cuda", 0)
if os.environ.get("HT_DEFAULT"):          # the reference's released default config
    cfg = get_cfg(num_layers=6, num_heads=10, units=500, inner_size=1000, tgt_length=128, mem_length=1024, batch_size=256,
                  batch_chunk=4, dropout=0.1, attention_dropout=0.1)
    T, Bt, WU = 128, 256, 14
else:
    cfg = get_cfg(num_layers=6, num_heads=8, units=512, inner_size=1024, tgt_length=1024, mem_length=0, batch_size=64,
                  batch_chunk=1, dropout=0.1, attention_dropout=0.1)
    T, Bt, WU = 1024, 64, 3
model = build_model(cfg, BaseVocab(), dev, seed=1); model.train()
tr = Trainer(model, cfg, num_gpus=1)
batches = [synthetic_batch(T, Bt, dev, seed=i) for i in range(4)]
for i in range(WU): tr.step(*batches[i % 4])
torch.cuda.synchronize()
N = 10
host = []
t0 = time.perf_counter()
for i in range(N):
    h0 = time.perf_counter()
    tr.step(*batches[i % 4])
    host.append(time.perf_counter() - h0)
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"host issue per step: {1e3*t_issue/N:.2f} ms (min {1e3*min(host):.2f}, max {1e3*max(host):.2f}); wall per step {1e3*t_all/N:.2f} ms")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for i in range(3): tr.step(*batches[i % 4])
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
