"""Phase timestamps of the decode layer-tail launch (100 MHz clock): mean over workgroups and launches."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "commu-code_amd"))
import ctypes as C
import numpy as np, torch
from commu_amd._lib import call
from commu_amd.generate import DecodeState
from commu_amd.model.config_helper import get_cfg
from commu_amd.model.dataset import BaseVocab
from commu_amd.train import build_model
dev = torch.device("cuda")
cfg = get_cfg(num_layers=6, num_heads=8, units=512, inner_size=1024, tgt_length=1, mem_length=4146, dropout=0.0,
              attention_dropout=0.0, same_length=True)
model = build_model(cfg, BaseVocab(), dev, seed=1).eval()
B = 64
st = DecodeState(model, B, 256)
st.prefill(torch.randint(2, 729, (11, B), device=dev))
tok = torch.randint(2, 729, (B,), device=dev)
ones = torch.ones(B, dtype=torch.uint8, device=dev)
for _ in range(5):
    st.step(tok, ones, ones)
buf = torch.zeros(128, 16, dtype=torch.int64, device=dev)
names = ["start", "P1 done", "wait1", "x+LN", "P2 done", "wait2", "P3 done", "wait3", "x+LN", "P4 done"]
acc = []
# trace ONE launch at a time: the last tail launch of a step (logits) overwrites the others, so trace layer by layer
import commu_amd.generate as G
orig = G.call
state = {"n": 0, "want": 0}
def hooked(name, *a):
    if name == "commu_decode_layer_tail":
        on = state["n"] == state["want"]
        orig("commu_decode_tail_trace", C.c_void_p(buf.data_ptr()) if on else None)
        state["n"] += 1
    r = orig(name, *a)
    if name == "commu_decode_layer_tail":
        orig("commu_decode_tail_trace", None)
    return r
G.call = hooked
for rep in range(20):
    state["n"], state["want"] = 0, rep % 5
    st.step(tok, ones, ones)
    torch.cuda.synchronize()
    t = buf.cpu().numpy()[:, :10].astype(np.float64)
    t0 = t[:, 0].min()
    acc.append((t - t0) / 100.0)          # us since the first workgroup started
a = np.stack(acc)                          # [rep, wg, stamp]
print("stamp            mean   min    max   (us since first workgroup start; over 128 workgroups x 20 launches)")
for i, n in enumerate(names):
    print(f"{n:12s} {a[:, :, i].mean():7.2f} {a[:, :, i].min():6.2f} {a[:, :, i].max():6.2f}")
