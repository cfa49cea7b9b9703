"""A/B of a PROCESS-WIDE switch inside the bench step (one trainer, the setting flipped between timing blocks, interleaved):
python ab_global.py kv3 | q3 | delta_in_kernel ...   ms per optimiser step at the bench shape (L6 D512 T1024 B64, dropout 0.1)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "commu-code_amd"))
import torch
from commu_amd import ops
from commu_amd.model.config_helper import get_cfg
from commu_amd.model.dataset import BaseVocab, synthetic_batch
from commu_amd.train import Trainer, build_model

SWITCH = {
    "kv3": lambda on: ops.attn_bwd_kv_generation(3 if on else 0),
    "q3": lambda on: ops.attn_bwd_kv_generation(4 if on else 0),
    "kv2": lambda on: ops.attn_bwd_kv_generation(2 if on else 0),
    "delta_in_kernel": lambda on: setattr(ops, "DELTA_KERNEL", not on),
    "fwd_saves_p": lambda on: setattr(ops, "FWD_SAVES_P", bool(on)),
    "tn8_96": lambda on: os.environ.__setitem__("COMMU_TN8_WGS", "96") if on else os.environ.pop("COMMU_TN8_WGS", None),
    "tn8_160": lambda on: os.environ.__setitem__("COMMU_TN8_WGS", "160") if on else os.environ.pop("COMMU_TN8_WGS", None),
    "tn8_192": lambda on: os.environ.__setitem__("COMMU_TN8_WGS", "192") if on else os.environ.pop("COMMU_TN8_WGS", None),
}


def env_switch(spec):          # "env:NAME=VALUE": an environment variable the library reads at every call
    name, val = spec[4:].split("=", 1)
    return lambda on: os.environ.__setitem__(name, val) if on else os.environ.pop(name, None)


for a in sys.argv[1:]:
    if a.startswith("env:"):
        SWITCH[a] = env_switch(a)
dev = torch.device("cuda", 0)
cfg = get_cfg()
model = build_model(cfg, BaseVocab(), dev, seed=1)
model.train()
tr = Trainer(model, cfg, num_gpus=1, settle_heap=False)
batches = [synthetic_batch(1024, 64, dev, seed=1111 + i) for i in range(4)]
names = sys.argv[1:] or ["kv3"]
for n in names:
    for on in (False, True):
        SWITCH[n](on)
        for i in range(4):
            tr.step(*batches[i % 4])
    SWITCH[n](False)
torch.cuda.synchronize()
for rnd in range(3):
    for n in ["base"] + names:
        if n != "base":
            SWITCH[n](True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(20):
            tr.step(*batches[i % 4])
        torch.cuda.synchronize()
        print(f"{n:20s} {1e3 * (time.perf_counter() - t0) / 20:.3f} ms/step", flush=True)
        if n != "base":
            SWITCH[n](False)
