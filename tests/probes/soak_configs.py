"""Soak over the shapes the bench line names, with the round's defaults (one-bit ReLU mask, relattn_kv3 as the key-stationary
kernel): many optimiser steps each, XL memory carried, resets on; finite and falling loss, no device fault."""
import math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "commu-code_amd"))
import torch
from commu_amd.model.config_helper import get_cfg
from commu_amd.model.dataset import BaseVocab, synthetic_batch
from commu_amd.train import Trainer, build_model
dev = torch.device("cuda")
CASES = [
    ("bench L6 D512 T1024 M0 B64", dict(num_layers=6, num_heads=8, units=512, inner_size=1024, tgt_length=1024, mem_length=0, batch_size=64, batch_chunk=1), 300, 0.0),
    ("mem1024", dict(num_layers=6, num_heads=8, units=512, inner_size=1024, tgt_length=1024, mem_length=1024, batch_size=64, batch_chunk=1), 60, 1.0),
    ("reference default D500 dh50 T128 M1024 B256/4", dict(num_layers=6, num_heads=10, units=500, inner_size=1000, tgt_length=128, mem_length=1024, batch_size=256, batch_chunk=4), 120, 0.25),
    ("cfg5 L12 D1024 T2048 M2048 B8", dict(num_layers=12, num_heads=16, units=1024, inner_size=2048, tgt_length=2048, mem_length=2048, batch_size=8, batch_chunk=1), 30, 1.0),
    ("odd: T96 M160 B12 H4", dict(num_layers=3, num_heads=4, units=256, inner_size=512, tgt_length=96, mem_length=160, batch_size=12, batch_chunk=3), 200, 0.3),
]
for name, kw, n, pr in CASES:
    cfg = get_cfg(dropout=0.1, attention_dropout=0.1, **kw)
    model = build_model(cfg, BaseVocab(), dev, seed=1)
    model.train()
    tr = Trainer(model, cfg)
    T, B = kw["tgt_length"], kw["batch_size"]
    batches = [synthetic_batch(T, B, dev, seed=100 + i, reset_prob=pr) for i in range(4)]
    t0 = time.time()
    first = last = None
    for s in range(n):
        loss = tr.step(*batches[s % 4])
        if s % 20 == 0 or s == n - 1:
            l = float(loss)
            assert math.isfinite(l), (name, s, l)
            first = l if first is None else first
            last = l
    torch.cuda.synchronize()
    print(f"{name}: {n} steps, loss {first:.3f} -> {last:.3f}, {1e3 * (time.time() - t0) / n:.2f} ms/step", flush=True)
    assert last < first
    del tr, model
    torch.cuda.empty_cache()
