"""Decode step of the wide model (L12 D1024 H16 DI2048, 64 sequences): layer-tail launches against the per-Linear chain."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "commu-code_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import commu_amd.generate as G
from test_configs_gpu import build
DEV = "cuda"
B = int(os.environ.get("DW_B", 64))
model, cfg, s, params = build(12, 16, 1024, 2048, 1, 4146, seed=23)
model.eval(); model.same_length = True; model.reset_length(1, 4146)
g = torch.Generator().manual_seed(5)
ctx = torch.randint(2, 729, (11, B), generator=g).to(DEV)
N = 300
toks = torch.randint(2, 729, (N, B), generator=g).to(DEV)
act = torch.ones(B, dtype=torch.uint8, device=DEV)
for tail in (True, False, True, False):
    st = G.DecodeState(model, B, 11 + N + 8)
    st.prefill(ctx)
    G.USE_LAYER_TAIL = tail
    for i in range(20):
        st.step(toks[i], act, act)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(20, N):
        st.step(toks[i], act, act)
    torch.cuda.synchronize()
    print(f"layer tail {tail}: {(time.perf_counter() - t0) / (N - 20) * 1e3:.3f} ms per step (eager launches, klen 31..{11 + N})", flush=True)
    del st
G.USE_LAYER_TAIL = True
