"""Soak of the in-launch hand-offs of the decode layer tail: N steps of the tail path against the per-Linear chain on the same
model / tokens, a second stream keeping the GPU unevenly busy; reports the worst logit difference and the give-up flag."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "commu-code_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import commu_amd.generate as G
from test_configs_gpu import build
N = int(os.environ.get("SOAK_STEPS", 4000))
dev = "cuda"
model, cfg, s, params = build(6, 8, 512, 1024, 1, 4146, seed=23)
model.eval(); model.same_length = True; model.reset_length(1, 4146)
g = torch.Generator().manual_seed(1)
B = 64
ctx = torch.randint(2, 729, (11, B), generator=g).to(dev)
st_a, st_b = G.DecodeState(model, B, 11 + N + 8), G.DecodeState(model, B, 11 + N + 8)
st_a.prefill(ctx); st_b.prefill(ctx)
side = torch.cuda.Stream()
big = torch.randn(3072, 3072, device=dev, dtype=torch.bfloat16)
worst, bad = 0.0, 0
ones = torch.ones(B, dtype=torch.uint8, device=dev)
for i in range(N):
    tok = torch.randint(2, 729, (B,), generator=g).to(dev)
    if i % 3 == 0:
        with torch.cuda.stream(side):
            for _ in range(1 + i % 4):
                big @ big
    la = st_a.step(tok, ones, ones).clone()
    G.USE_LAYER_TAIL = False
    lb = st_b.step(tok, ones, ones).clone()
    G.USE_LAYER_TAIL = True
    if i % 50 == 49 or i == N - 1:
        err = float((la[:, :729] - lb[:, :729]).abs().max()) / float(lb[:, :729].abs().max())
        worst = max(worst, err)
        bad += err > 1e-2
torch.cuda.synchronize()
st_a.check()
print(f"{N} steps (klen 11 -> {11 + N}), 64 sequences, loaded: worst logit difference {worst:.2e} of range over the sampled steps, "
      f"{bad} above 1e-2, no hand-off gave up")
