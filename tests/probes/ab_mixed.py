"""A/B of model attributes AND environment switches inside the bench step, one process, interleaved:
python ab_mixed.py attr:resid_in_ln_bwd env:COMMU_GEMM8_SKEW=2000 both:resid_in_ln_bwd+COMMU_GEMM8_SKEW=2000"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "commu-code_amd"))
import torch
from commu_amd.model.config_helper import get_cfg
from commu_amd.model.dataset import BaseVocab, synthetic_batch
from commu_amd.train import Trainer, build_model
dev = torch.device("cuda", 0)
cfg = get_cfg()
model = build_model(cfg, BaseVocab(), dev, seed=1)
model.train()
tr = Trainer(model, cfg, num_gpus=1, settle_heap=False)
batches = [synthetic_batch(1024, 64, dev, seed=1111 + i) for i in range(4)]


def apply(spec, on):
    kind, body = spec.split(":", 1)
    for item in body.split("+"):
        if "=" in item:
            name, val = item.split("=", 1)
            if on:
                os.environ[name] = val
            else:
                os.environ.pop(name, None)
        else:
            setattr(model, item, bool(on))


specs = sys.argv[1:]
for sp in specs:
    apply(sp, True)
    for i in range(4):
        tr.step(*batches[i % 4])
    apply(sp, False)
for i in range(4):
    tr.step(*batches[i % 4])
torch.cuda.synchronize()
for rnd in range(4):
    for sp in ["base"] + specs:
        if sp != "base":
            apply(sp, True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(20):
            tr.step(*batches[i % 4])
        torch.cuda.synchronize()
        print(f"{sp:60s} {1e3 * (time.perf_counter() - t0) / 20:.3f} ms/step", flush=True)
        if sp != "base":
            apply(sp, False)
