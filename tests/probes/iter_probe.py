"""Where does the iterator-fed training loop wait?  (probe)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "commu-code_amd"))
import numpy as np, torch
import commu_amd.model.dataset as D
from commu_amd.model.config_helper import get_cfg
from commu_amd.model.dataset import BaseVocab
from commu_amd.train import Trainer, build_model
dev = torch.device("cuda", 0)
T, B = 1024, 64
rng = np.random.default_rng(1111)
seqs = [np.concatenate([rng.integers(560, 729, size=11), rng.integers(2, 560, size=int(n))]) for n in rng.integers(T // 2, 3 * T, size=600)]
cfg = get_cfg(num_layers=6, num_heads=8, units=512, inner_size=1024, tgt_length=T, mem_length=0, batch_size=B, batch_chunk=1, dropout=0.1, attention_dropout=0.1)
ds = D.ComMUDataset(None, cfg, sequences={"train": seqs, "valid": seqs[:70]})
model = build_model(cfg, BaseVocab(), dev, seed=1); model.train()
tr = Trainer(model, cfg)
stats = {"get": 0.0, "wev": 0.0, "produce": 0.0, "to": 0.0, "sync": 0.0, "put": 0.0}
P = D._Prefetcher
orig_next = P.__next__
def timed_next(self):
    t0 = time.perf_counter(); item = self.ready.get(); t1 = time.perf_counter()
    stats["get"] += t1 - t0
    if item is None: raise StopIteration
    out, ev, extra = item
    if ev is not None:
        cur = torch.cuda.current_stream(self.device); cur.wait_event(ev)
        for t in out: t.record_stream(cur)
    stats["wev"] += time.perf_counter() - t1
    return out, extra
P.__next__ = timed_next
orig_gather = D.gather_batch
def tg(*a):
    t0 = time.perf_counter(); r = orig_gather(*a); stats["produce"] += time.perf_counter() - t0; return r
D.gather_batch = tg
it = ds.get_iterator(B, T, dev, "train", True, seed=1)()
for _ in range(5): tr.step(*next(it))
torch.cuda.synchronize()
for k in stats: stats[k] = 0.0
N = 20
t0 = time.perf_counter(); tstep = 0.0
for _ in range(N):
    b = next(it)
    t1 = time.perf_counter(); tr.step(*b); tstep += time.perf_counter() - t1
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"ms/step {1e3*dt/N:.2f}; host in step {1e3*tstep/N:.2f}; " + ", ".join(f"{k} {1e3*v/N:.2f}" for k, v in stats.items()))
# the same loop with the batches made resident first (same data path otherwise)
bs = [next(it) for _ in range(4)]
torch.cuda.synchronize()
t0 = time.perf_counter(); tstep = 0.0
for i in range(N):
    t1 = time.perf_counter(); tr.step(*bs[i % 4]); tstep += time.perf_counter() - t1
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"resident: ms/step {1e3*dt/N:.2f}; host in step {1e3*tstep/N:.2f}")
it.close()
from commu_amd.model.dataset import synthetic_batch
sb = [synthetic_batch(T, B, dev, seed=i) for i in range(4)]
for name, batches in (("synthetic", sb), ("iterator-made", bs), ("synthetic", sb), ("iterator-made", bs)):
    for i in range(3): tr.step(*batches[i % 4])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(N): tr.step(*batches[i % 4])
    torch.cuda.synchronize()
    print(f"{name}: {1e3*(time.perf_counter()-t0)/N:.2f} ms/step")
from commu_amd import _lib
names = ["commu_embed_bwd_sorted", "commu_embed_fwd", "commu_ce_fwd", "commu_ce_bwd", "commu_relattn_fwd", "commu_relattn_bwd_q", "commu_relattn_bwd_kv", "commu_gemm_nt_bf16", "commu_layernorm_bwd", "commu_reduce_slabs_f32"]
for name, batches in (("synthetic", sb), ("iterator-made", bs)):
    _lib.profile_start(names)
    for i in range(8): tr.step(*batches[i % 4])
    pr = _lib.profile_stop()
    print(name, {k.replace("commu_", ""): round(sum(v) / 8, 3) for k, v in pr.items() if v})
