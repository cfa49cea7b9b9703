"""Our GEMMs vs the vendor library (torch -> hipBLASLt/rocBLAS) at the step's shapes: what is attainable."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "commu-code_amd"))
import torch
from commu_amd import ops
def t(f, n=20):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for (M, N, K) in [(65536, 1536, 512), (65536, 512, 512), (65536, 1024, 512), (65536, 512, 1024), (65536, 768, 512), (65536, 512, 1536)]:
    A = torch.randn(M, K, device="cuda").bfloat16(); W = torch.randn(N, K, device="cuda").bfloat16()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    us = t(lambda: ops.gemm_nt(A, W, out=out))
    ub = t(lambda: torch.matmul(A, W.t(), out=out))
    print(f"NT {M}x{N}x{K}: ours {us:8.1f} us {2*M*N*K/us/1e6:6.0f} TF | blas {ub:8.1f} us {2*M*N*K/ub/1e6:6.0f} TF")
    del A, W, out
# TN (weight gradients): C[N1, N2] = A[M, N1]^T B[M, N2], fp32 out
for (M, N1, N2) in [(65536, 1536, 512), (65536, 512, 512), (65536, 1024, 512), (65536, 512, 1024)]:
    A = torch.randn(M, N1, device="cuda").bfloat16(); Bm = torch.randn(M, N2, device="cuda").bfloat16()
    acc = torch.zeros(N1, N2, device="cuda")
    us = t(lambda: ops.gemm_tn(A, Bm, acc))
    outb = torch.empty(N1, N2, device="cuda", dtype=torch.bfloat16)
    ub = t(lambda: torch.matmul(A.t(), Bm, out=outb))
    print(f"TN {N1}x{N2}x{M}: ours {us:8.1f} us {2*M*N1*N2/us/1e6:6.0f} TF | blas {ub:8.1f} us {2*M*N1*N2/ub/1e6:6.0f} TF")
