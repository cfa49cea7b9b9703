"""gemm_nt8 with the pipelined epilogue: start-up skew between workgroups (COMMU_GEMM8_SKEW cycles x 8 phase groups)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "commu-code_amd"))
from commu_amd import ops
dev = torch.device("cuda:0")
M = 65536


def timeit(fn, n=30):
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for N, K in ((1024, 512), (512, 512), (1536, 512), (512, 1024)):
    x = torch.randn(M, K, device=dev).bfloat16()
    w = torch.randn(N, K, device=dev).bfloat16()
    y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    row = []
    for pipe in (False, True):
        for skew in (0, 500, 1000, 2000, 4000):
            os.environ["COMMU_GEMM8_NOPIPE"] = "1"
            if pipe:
                os.environ.pop("COMMU_GEMM8_NOPIPE", None)
            os.environ["COMMU_GEMM8_SKEW"] = str(skew)
            row.append(f"{'P' if pipe else 'B'}{skew}:{timeit(lambda: ops.gemm_nt(x, w, out=y)):6.1f}")
    print(f"({M},{N},{K})  " + "  ".join(row))
