import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "commu-code_amd"))
import torch
from commu_amd import ops
def t(f, n=50):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for (M, N, K) in [(64, 1536, 512), (64, 512, 512), (64, 1024, 512), (64, 512, 1024), (64, 729, 512), (3, 1536, 512)]:
    A = torch.randn(M, K, device="cuda").bfloat16(); W = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    bias = torch.randn(N, device="cuda"); res = torch.randn(M, N, device="cuda").bfloat16()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    ref = torch.relu(A.float() @ W.float().t() + bias) + res.float()
    ops.gemm_nt(A, W, out=out, bias=bias, relu=True, resid=res)
    err = float((out.float() - ref).abs().max() / ref.abs().max())
    us = t(lambda: ops.gemm_nt(A, W, out=out, bias=bias, relu=True, resid=res))
    os.environ["COMMU_GEMM_NOSKINNY"] = "1"
    us0 = t(lambda: ops.gemm_nt(A, W, out=out, bias=bias, relu=True, resid=res))
    del os.environ["COMMU_GEMM_NOSKINNY"]
    print(f"{M}x{N}x{K}: skinny {us:6.1f} us  tiled {us0:6.1f} us  relerr {err:.4f}")
