"""Step time of Trainer.step: eager vs hipGraph replay, with / without the side streams (probe)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "commu-code_amd"))
import torch
from commu_amd.model.config_helper import get_cfg
from commu_amd.model.dataset import BaseVocab, synthetic_batch
from commu_amd.train import Trainer, build_model

dev = torch.device("cuda", 0)
B = int(os.environ.get("GP_B", 64))
T = int(os.environ.get("GP_T", 1024))
STEPS = int(os.environ.get("GP_STEPS", 20))


def run(graph, side):
    cfg = get_cfg(num_layers=6, num_heads=8, units=512, inner_size=1024, tgt_length=T, mem_length=0, batch_size=B,
                  batch_chunk=1, dropout=0.1, attention_dropout=0.1)
    model = build_model(cfg, BaseVocab(), dev, seed=1)
    model.train()
    model.wgrad_side_stream = side
    tr = Trainer(model, cfg, graph=graph)
    bs = [synthetic_batch(T, B, dev, seed=i) for i in range(4)]
    for i in range(6):
        tr.step(*bs[i % 4])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(STEPS):
        tr.step(*bs[i % 4])
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"B={B} graph={graph} side={side}: {1e3*dt/STEPS:.3f} ms/step (host issue {1e3*th/STEPS:.3f} ms) failed={tr.graph_failed}", flush=True)
    del tr, model
    torch.cuda.empty_cache()


for rep in range(2):
    for graph in (False, True):
        for side in (True, False):
            run(graph, side)
