"""gemm8 timing at the step's NT shapes + long-K shapes; env sweep COMMU_GEMM8_SKEW (the COMMU_GEMM8_ABL ablation switch this probe
also swept was removed from the library: the ablated kernels compute wrong results and are not reachable from the product path)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "commu-code_amd"))
import torch
from commu_amd import ops
def t(f, n=20):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
VAR = os.environ.get("G8_SWEEP", "COMMU_GEMM8_ABL")
VALS = os.environ.get("G8_VALS", "0,1,2,3").split(",")
shapes = [(65536, 1536, 512), (65536, 512, 512), (65536, 1024, 512), (65536, 512, 1024), (65536, 768, 512), (65536, 512, 1536), (65536, 1536, 4096)]
for (M, N, K) in shapes:
    A = torch.randn(M, K, device="cuda").bfloat16(); W = torch.randn(N, K, device="cuda").bfloat16()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    row = f"NT {M}x{N}x{K}:"
    for v in VALS:
        os.environ[VAR] = v
        us = t(lambda: ops.gemm_nt(A, W, out=out))
        row += f"  {v}: {us:7.1f} us {2*M*N*K/us/1e6:5.0f} TF |"
    os.environ[VAR] = "0"
    print(row, flush=True)
    del A, W, out
