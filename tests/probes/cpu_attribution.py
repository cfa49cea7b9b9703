"""Build-container only: speed of the oracle's train / generation step vs the imported reference on the same threads
('attribution check'): the CPU baseline bench.py times on the GPU node is the oracle
(kind "port"); this ratio says how far it is from the reference's own code.  Needs /root/reference."""
import os, sys, time, types
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, "/root/reference")
import torch
torch.set_num_threads(8)
from commu.model.model import MemTransformerLM
from oracle import xl_ref as X
ns = lambda **k: types.SimpleNamespace(**k)
L, H, D, DI, T, B = 6, 8, 512, 1024, 1024, 2
cfg = ns(MODEL=ns(num_layers=L, num_heads=H, units=D, inner_size=DI, dropout=0.0, attention_dropout=0.0, same_length=False, clamp_len=-1),
         TRAIN=ns(tgt_length=T, mem_length=0))
class V:
    pad_id = 0
    def __len__(self): return 729
torch.manual_seed(0)
ref = MemTransformerLM(cfg, V())
for p in ref.parameters(): torch.nn.init.normal_(p, 0, 0.02)
opt = torch.optim.Adam(ref.parameters(), lr=1e-4)
g = torch.Generator().manual_seed(0)
def ref_step():
    s = torch.randint(2, 729, (T + 1, B), generator=g)
    ref.zero_grad()
    loss, _ = ref(s[:-1], s[1:], torch.zeros(B, dtype=torch.bool), None)
    loss[s[1:] != 0].float().mean().backward()
    torch.nn.utils.clip_grad_norm_(ref.parameters(), 1.0)
    opt.step()
sh = X.XLShape(L, H, D, DI); p = X.init_params(sh, 1); st = X.adam_init(p)
def ora_step():
    s = torch.randint(2, 729, (T + 1, B), generator=g)
    X.train_step(p, st, sh, s[:-1], s[1:], torch.zeros(B, dtype=torch.bool), [None], batch_chunk=1, mem_len=0, same_length=False, lr_now=1e-4, clip=1.0)
def t(f, n=2):
    f(); t0 = time.time()
    for _ in range(n): f()
    return (time.time() - t0) / n
tr, to = t(ref_step), t(ora_step)
print(f"train step L{L} D{D} T{T} B{B}, 8 threads: reference {tr:.2f} s ({B*T/tr:.0f} tok/s), oracle {to:.2f} s ({B*T/to:.0f} tok/s), oracle/reference time = {to/tr:.2f}")
# generation step at memory length 1000 (batch 1)
ref.eval(); ref.same_length = True; ref.reset_length(1, 4146)
with torch.no_grad():
    ctx = torch.randint(2, 729, (1000, 1), generator=g)
    _, rm = ref.forward_generate(ctx, None)
    _, om = X.forward_generate(p, sh, ctx, None, 4146)
    tok = torch.tensor([[5]])
    def rg():
        global rm
        _, rm = ref.forward_generate(tok, rm)
    def og():
        global om
        _, om = X.forward_generate(p, sh, tok, om, 4146)
    gr, go = t(rg, 10), t(og, 10)
print(f"generation step, memory 1000, batch 1: reference {gr*1e3:.1f} ms, oracle {go*1e3:.1f} ms, oracle/reference time = {go/gr:.2f}")
