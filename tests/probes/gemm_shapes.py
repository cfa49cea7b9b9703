import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "commu-code_amd"))
import torch
from commu_amd import ops
def t(f, n=10):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for (M, N, K) in [(8192, 8192, 8192), (4096, 4096, 4096), (65536, 1536, 512), (65536, 1536, 2048), (65536, 1536, 8192), (65536, 512, 512), (65536, 512, 4096), (16384, 1536, 512)]:
    A = torch.randn(M, K, device="cuda").bfloat16(); W = torch.randn(N, K, device="cuda").bfloat16()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    us = t(lambda: ops.gemm_nt(A, W, out=out))
    print(f"NT {M}x{N}x{K}: {us:9.1f} us  {2*M*N*K/us/1e6:7.0f} TF")
    del A, W, out
