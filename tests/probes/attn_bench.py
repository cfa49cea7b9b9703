"""Micro-benchmark of the attention kernels at the bench shape (B x H x T, M = 0) for profiling."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "commu-code_amd"))
import torch  # noqa: E402
from commu_amd import ops  # noqa: E402
ops.attn_fwd_generation(int(os.environ.get("COMMU_ATTN_FWD_GEN", "0")))      # 2: the 16x16-layout forward
ops.attn_bwd_kv_generation(int(os.environ.get("COMMU_ATTN_KV_GEN", "0")))     # 2 / 3: key-stationary backward kernel

B = int(os.environ.get("AB_B", 16))
T = int(os.environ.get("AB_T", 1024))
M = int(os.environ.get("AB_M", 0))
H, DH = int(os.environ.get("AB_H", 8)), 64
REPS = int(os.environ.get("AB_REPS", 3))
WHAT = os.environ.get("AB_WHAT", "fwd,bwd")
DROP = float(os.environ.get("AB_DROP", 0.0))
ops.FWD_SAVES_P = os.environ.get("AB_SAVEP", "0") != "0"          # forward-saved probabilities (opt-in; the training default is off)
dev = "cuda"
K = T + M
HD = H * DH
g = torch.Generator().manual_seed(0)
qkv = (torch.randn(K * B, 3 * HD, generator=g) * 0.7).to(torch.bfloat16).to(dev)
rd = (torch.randn(K, HD, generator=g) * 0.7).to(torch.bfloat16).to(dev)
u = (torch.randn(HD, generator=g) * 0.3).to(dev)
vb = (torch.randn(HD, generator=g) * 0.3).to(dev)
dout = (torch.randn(T * B, HD, generator=g)).to(torch.bfloat16).to(dev)
q, k, v = qkv[M * B:, :HD], qkv[:, HD:2 * HD], qkv[:, 2 * HD:]
dqkv = torch.zeros_like(qkv)
drd = torch.zeros(K, HD, device=dev)
du, dvb = torch.zeros(HD, device=dev), torch.zeros(HD, device=dev)


SCR = {} if os.environ.get("AB_SCRATCH", "1") != "0" else None          # persistent zero-initialised dS scratch, like the model


def run():
    out, lse, qs = ops.relattn_fwd(q, k, v, rd, u, vb, None, T, M, B, H, DH, False, M, save_q=True, save_p=True, drop_p=DROP, drop_seed=1234)
    if "bwd" in WHAT:
        ops.relattn_bwd(q, k, v, rd, u, vb, None, T, M, B, H, DH, False, M, out, dout, lse, qs, dqkv[M * B:, :HD],
                        dqkv[:, HD:2 * HD], dqkv[:, 2 * HD:], drd, du, dvb, drop_p=DROP, drop_seed=1234, scratch=SCR)


run()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(REPS):
    run()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / REPS
kbar = M + (T + 1) / 2
fl = B * T * 6 * kbar * HD
print(f"B={B} T={T} M={M}: {dt*1e3:.3f} ms per fwd{'+bwd' if 'bwd' in WHAT else ''}; fwd algorithmic {fl/1e9:.1f} GF")
