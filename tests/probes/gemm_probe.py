import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "commu-code_amd"))
import torch
from commu_amd import ops
M, N, K = [int(x) for x in os.environ.get("GP_SHAPE", "65536,1536,512").split(",")]
A = torch.randn(M, K, device="cuda").bfloat16(); W = torch.randn(N, K, device="cuda").bfloat16()
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
for _ in range(int(os.environ.get("GP_REPS", 5))):
    ops.gemm_nt(A, W, out=out)
torch.cuda.synchronize()
