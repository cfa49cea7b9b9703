"""Host profile of the eager step at 8 sequences per GPU (where the host is the bound)."""
import cProfile, os, pstats, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "commu-code_amd"))
from commu_amd.model.config_helper import get_cfg
from commu_amd.model.dataset import BaseVocab, synthetic_batch
from commu_amd.train import Trainer, build_model
dev = torch.device("cuda")
cfg = get_cfg(num_layers=6, num_heads=8, units=512, inner_size=1024, tgt_length=1024, mem_length=0, batch_size=8, batch_chunk=1,
              dropout=0.1, attention_dropout=0.1)
model = build_model(cfg, BaseVocab(), dev, seed=3)
model.train()
tr = Trainer(model, cfg, num_gpus=1, graph=False)
d, t, r, n = synthetic_batch(1024, 8, dev, seed=9)
for _ in range(6):
    tr.step(d, t, r, n)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    tr.step(d, t, r, n)
th = time.perf_counter() - t0
torch.cuda.synchronize()
print("ms per step", (time.perf_counter() - t0) / 20 * 1e3, " host only", th / 20 * 1e3)
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    tr.step(d, t, r, n)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
