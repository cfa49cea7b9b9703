"""Attribution of the NT GEMM epilogue cost at the step's shapes (isolated)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "commu-code_amd"))
import torch
from commu_amd import ops

def t(f, n=30):
    f(); f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6

M = 65536
for N, K in ((512, 512), (512, 1024), (1024, 512), (1536, 512), (512, 1536)):
    A = torch.randn(M, K, device="cuda").bfloat16(); W = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    bias = torch.randn(N, device="cuda"); resid = torch.randn(M, N, device="cuda").bfloat16()
    row = f"N={N} K={K}:"
    for name, kw in (("plain", {}), ("bias", dict(bias=bias)), ("drop", dict(drop_p=0.1, drop_seed=5)), ("resid", dict(resid=resid)),
                     ("resid+drop", dict(resid=resid, drop_p=0.1, drop_seed=5)), ("relu_mask", dict(relu_mask=resid, mask_scale=1.1)),
                     ("bias+relu+drop", dict(bias=bias, relu=True, drop_p=0.1, drop_seed=5))):
        us = t(lambda: ops.gemm_nt(A, W, out=out, **kw))
        row += f"  {name} {us:6.1f}"
    ref = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    row += f"  | blas {t(lambda: torch.matmul(A, W.t(), out=ref)):6.1f}"
    # what a separate residual-add pass costs (read 2, write 1)
    row += f"  | torch add_ {t(lambda: out.add_(resid)):6.1f}"
    print(row, flush=True)
