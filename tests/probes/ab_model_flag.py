"""A/B of a model attribute inside the bench step (same process, interleaved): python ab_model_flag.py <attr> [<attr> ...]
e.g. relu_sign_bits.  ms per optimiser step at the bench shape (L6 D512 T1024 B64, dropout 0.1), 3 x 20 steps per setting."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "commu-code_amd"))
import torch
from commu_amd.model.config_helper import get_cfg
from commu_amd.model.dataset import BaseVocab, synthetic_batch
from commu_amd.train import Trainer, build_model

dev = torch.device("cuda", 0)
cfg = get_cfg()
attrs = sys.argv[1:] or ["relu_sign_bits"]


def make(flags):
    model = build_model(cfg, BaseVocab(), dev, seed=1)
    model.train()
    for a in flags:
        setattr(model, a, True)
    tr = Trainer(model, cfg, num_gpus=1, settle_heap=False)
    return tr


batches = [synthetic_batch(1024, 64, dev, seed=1111 + i) for i in range(4)]
trainers = {"base": make([])}
for a in attrs:
    trainers[a] = make([a])
for tr in trainers.values():
    for i in range(6):
        tr.step(*batches[i % 4])
torch.cuda.synchronize()
for rnd in range(3):
    for name, tr in trainers.items():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(20):
            tr.step(*batches[i % 4])
        torch.cuda.synchronize()
        print(f"{name:20s} {1e3 * (time.perf_counter() - t0) / 20:.3f} ms/step", flush=True)
