"""MX-fp8 NT GEMM vs the bf16 path at the model's GEMM shapes (isolated, TFLOP/s)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "commu-code_amd"))
import torch
from commu_amd import ops
dev = "cuda"
def t(f, n=20):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
for M, N, K in [(65536, 1536, 512), (65536, 1024, 512), (65536, 512, 1024), (16384, 3072, 1024), (16384, 2048, 1024), (16384, 1024, 2048), (8192, 8192, 8192)]:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16); b = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
    qa, sa = ops.quant_mxfp8(a); qb, sb = ops.quant_mxfp8(b)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    tq = t(lambda: ops.quant_mxfp8(a))
    t8 = t(lambda: ops.gemm_nt_mxfp8(qa, sa, qb, sb, out=out))
    tb = t(lambda: ops.gemm_nt(a, b, out=out))
    fl = 2.0 * M * N * K
    print(f"M{M} N{N} K{K}: mxfp8 {t8*1e6:.0f} us ({fl/t8/1e12:.0f} TF/s) + quant(A) {tq*1e6:.0f} us | bf16 {tb*1e6:.0f} us ({fl/tb/1e12:.0f} TF/s)")
