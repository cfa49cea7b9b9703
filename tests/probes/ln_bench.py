import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "commu-code_amd"))
import torch
from commu_amd import ops
rows, D = 65536, 512
z = torch.randn(rows, D, device="cuda").bfloat16()
g = torch.ones(D, device="cuda"); b = torch.zeros(D, device="cuda")
y = torch.empty_like(z); yd = torch.empty_like(z)
mean = torch.empty(rows, device="cuda"); rstd = torch.empty(rows, device="cuda")
def t(f, n=20):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
print("ln fwd plain      %.1f us" % t(lambda: ops.layernorm_fwd(z, g, b, y, mean, rstd)))
print("ln fwd + y_drop   %.1f us" % t(lambda: ops.layernorm_fwd(z, g, b, y, mean, rstd, y_drop=yd, drop_p=0.1, drop_seed=5)))
dz = torch.empty_like(z); dzm = torch.empty_like(z)
print("ln bwd plain      %.1f us" % t(lambda: ops.layernorm_bwd(z, z, mean, rstd, g, dz)))
print("ln bwd + masked   %.1f us" % t(lambda: ops.layernorm_bwd(z, z, mean, rstd, g, dz, dz_masked=dzm, drop_p=0.1, drop_seed=5)))
A = torch.randn(65536, 512, device="cuda").bfloat16(); W = torch.randn(1536, 512, device="cuda").bfloat16()
out = torch.empty(65536, 1536, device="cuda", dtype=torch.bfloat16)
us = t(lambda: ops.gemm_nt(A, W, out=out)); print("gemm qkv plain    %.1f us  %.0f TF" % (us, 2*65536*512*1536/us/1e6))
us = t(lambda: ops.gemm_nt(A, W, out=out, drop_p=0.1, drop_seed=3)); print("gemm qkv dropout  %.1f us  %.0f TF" % (us, 2*65536*512*1536/us/1e6))
W2 = torch.randn(512, 1024, device="cuda").bfloat16(); A2 = torch.randn(65536, 1024, device="cuda").bfloat16()
out2 = torch.empty(65536, 512, device="cuda", dtype=torch.bfloat16)
us = t(lambda: ops.gemm_nt(A2, W2, out=out2, resid=out2)); print("gemm ffn2 +resid   %.1f us  %.0f TF" % (us, 2*65536*512*1024/us/1e6))
Wo = torch.randn(512, 512, device="cuda").bfloat16()
us = t(lambda: ops.gemm_nt(A, Wo, out=out2)); print("gemm o_net        %.1f us  %.0f TF" % (us, 2*65536*512*512/us/1e6))
g32 = torch.zeros(1536, 512, device="cuda")
us = t(lambda: ops.gemm_tn(out, A, g32)); print("gemm_tn dWqkv     %.1f us  %.0f TF" % (us, 2*65536*512*1536/us/1e6))
X5 = torch.randn(65536, 512, device="cuda").bfloat16(); X10 = torch.randn(65536, 1024, device="cuda").bfloat16()
o5 = torch.zeros(512, device="cuda"); o10 = torch.zeros(1024, device="cuda")
print("colsum 512        %.1f us" % t(lambda: ops.colsum(X5, o5)))
print("colsum 1024       %.1f us" % t(lambda: ops.colsum(X10, o10)))
