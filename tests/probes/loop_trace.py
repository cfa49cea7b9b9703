"""Stage timestamps of the fused sampling-step / post / pre launch of the decode loop (100 MHz clock)."""
import os, sys, types
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "commu-code_amd"))
import ctypes as C
import numpy as np, torch
from commu_amd._lib import call
from commu_amd.generate import ForcedDecoder
from commu_amd.model.config_helper import get_cfg
from commu_amd.model.dataset import BaseVocab
from commu_amd.train import build_model
dev = torch.device("cuda")
cfg = get_cfg(num_layers=6, num_heads=8, units=512, inner_size=1024, tgt_length=1, mem_length=4146, dropout=0.0,
              attention_dropout=0.0, same_length=True)
model = build_model(cfg, BaseVocab(), dev, seed=1).eval()
B = 64
with torch.no_grad():
    bias = model.crit.out_layers[0].bias
    bias.zero_(); bias[1:3] = -1e9; bias[195:304] = -1e9
    dec = ForcedDecoder(model, B, generation_length=256, memory_length=4146, temperature=0.95, top_k=32)
    data = types.SimpleNamespace(num_measures=4.0, chord_token_components={"chord_token": [], "chord_position": []})
    meta = [574, 623, 627, 635, 639, 642, 651, 684, 694, 720, 727]
    dec.load([meta] * B, [data] * B, torch.rand(B, dec.ld_u).numpy())
    dec.pre()
    for _ in range(8):
        dec.body_pre()
    buf = torch.zeros(B, 4, dtype=torch.int64, device=dev)
    call("commu_decode_loop_trace", C.c_void_p(buf.data_ptr()))
    acc = []
    for _ in range(20):
        dec.body_pre()
        torch.cuda.synchronize()
        t = buf.cpu().numpy().astype(np.float64)
        acc.append((t - t[:, :1].min()) / 100.0)
    call("commu_decode_loop_trace", None)
a = np.stack(acc)
for i, n in enumerate(["start", "sampled", "post done", "pre done"]):
    print(f"{n:10s} mean {a[:, :, i].mean():6.2f}  min {a[:, :, i].min():6.2f}  max {a[:, :, i].max():6.2f} us")
