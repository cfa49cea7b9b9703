"""Per-shape table of the GEMM launches of the REAL bench step (HIP events around every call, side streams running):
shape, launches per step, mean time, TFLOP/s.  Output: gpurun_out/r02_gemm_shapes.txt (copy to profiles/)."""
import ctypes as C, os, sys
from collections import defaultdict
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "commu-code_amd")); sys.path.insert(0, ROOT)
import torch
from commu_amd import _lib
from commu_amd.model.config_helper import get_cfg
from commu_amd.model.dataset import BaseVocab, synthetic_batch
from commu_amd.train import Trainer, build_model
dev = torch.device("cuda", 0)
cfg = get_cfg(num_layers=6, num_heads=8, units=512, inner_size=1024, tgt_length=1024, mem_length=0, batch_size=64,
              batch_chunk=1, dropout=0.1, attention_dropout=0.1)
model = build_model(cfg, BaseVocab(), dev, seed=1); model.train()
tr = Trainer(model, cfg, num_gpus=1)
batches = [synthetic_batch(1024, 64, dev, seed=i) for i in range(4)]
for i in range(3): tr.step(*batches[i % 4])
torch.cuda.synchronize()
rec = []
orig = _lib.call
def val(x):
    return x.value if hasattr(x, "value") else x
def call(name, *args):
    if name in ("commu_gemm_nt_bf16", "commu_gemm_tn_bf16_grouped", "commu_gemm_tn_bf16"):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); rc = orig(name, *args); e.record()
        if name == "commu_gemm_nt_bf16":
            key, fl = ("NT", val(args[6]), val(args[7]), val(args[8])), 2.0 * val(args[6]) * val(args[7]) * val(args[8])
        elif name == "commu_gemm_tn_bf16":
            M, N, K = val(args[7]), val(args[8]), val(args[9])
            key, fl = ("TN", M, N, K), 2.0 * M * N * K
        else:
            probs, n, M = args[0], val(args[1]), val(args[2])
            shapes = tuple((probs[i].N, probs[i].K) for i in range(n))
            key, fl = ("TN grouped", M, shapes), sum(2.0 * M * a * b for a, b in shapes)
        rec.append((key, fl, s, e))
        return rc
    return orig(name, *args)
_lib.call = call
import commu_amd.ops as ops
ops.call = call
N = 6
for i in range(N): tr.step(*batches[i % 4])
torch.cuda.synchronize()
agg = defaultdict(lambda: [0, 0.0, 0.0])
for key, fl, s, e in rec:
    a = agg[key]; a[0] += 1; a[1] += s.elapsed_time(e); a[2] = fl
lines = ["# GEMM launches of the bench step (L6 D512 H8 DI1024 T1024 B64, dropout 0.1), HIP events around each call inside",
         "# Trainer.step with the side streams running; M = tokens (NT: rows of the output; TN: contraction length)",
         f"# {'kind':<11}{'shape':<52}{'calls/step':>10}{'mean us':>10}{'TFLOP/s':>9}"]
tot = 0.0
for key, (n, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    us = 1e3 * ms / n
    tot += ms / N
    lines.append(f"  {key[0]:<11}{str(key[1:]):<52}{n / N:>10.1f}{us:>10.1f}{fl / us / 1e6:>9.0f}")
lines.append(f"# sum over shapes: {tot:.2f} ms per step")
out = os.path.join(ROOT, "gpurun_out", "r02_gemm_shapes.txt")
open(out, "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
