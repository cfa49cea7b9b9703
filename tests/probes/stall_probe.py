"""Per-step wall times of the released-default config in the order the bench's extra rows run it (fresh trainer each:
no resets, resets, micro-batch loop), several rounds: looks for intermittent multi-second stalls."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "commu-code_amd"))
import torch
from commu_amd.model.config_helper import get_cfg
from commu_amd.model.dataset import BaseVocab, synthetic_batch
from commu_amd.train import Trainer, build_model
dev = torch.device("cuda")
for rep in range(int(os.environ.get("REPS", 4))):
    for tag, pr, merge in (("plain", 0.0, None), ("resets", 0.25, None), ("loop", 0.0, False)):
        cfg = get_cfg(num_layers=6, num_heads=10, units=500, inner_size=1000, tgt_length=128, mem_length=1024, batch_size=256,
                      batch_chunk=4, dropout=0.1, attention_dropout=0.1)
        model = build_model(cfg, BaseVocab(), dev, seed=1).train()
        tr = Trainer(model, cfg, merge_chunks=merge)
        batches = [synthetic_batch(128, 256, dev, seed=1111 + i, reset_prob=pr) for i in range(4)]
        times = []
        for i in range(22):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            tr.step(*batches[i % 4])
            torch.cuda.synchronize(); times.append(1e3 * (time.perf_counter() - t0))
        tail = times[14:]
        print(f"rep {rep} {tag:6s}: timed steps {[round(t, 1) for t in tail]}", flush=True)
        del tr, model, batches
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
