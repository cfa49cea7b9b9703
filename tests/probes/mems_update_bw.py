"""Bandwidth of commu_mems_update at the released default configuration's shape (7 layers x 1024 x 64 x 512 bf16)."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "commu-code_amd"))
from commu_amd import ops
L, M, T, B, D = 7, 1024, 128, 64, 512
hids = torch.randn(L, T * B, D, device="cuda").bfloat16()
mems = torch.randn(L, M, B, D, device="cuda").bfloat16()
out = torch.empty(L, M, B, D, device="cuda", dtype=torch.bfloat16)
for _ in range(3):
    ops.mems_update(hids, mems, out, T)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    ops.mems_update(hids, mems, out, T)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 20
ref = torch.cat([mems, hids.view(L, T, B, D)], 1)[:, T:]
print(f"{dt * 1e6:.1f} us per call, {2 * out.numel() * 2 / dt / 1e12:.2f} TB/s, exact {torch.equal(out, ref)}")
