"""The step's NT GEMM shapes, one launch each, isolated (for a PMC pass: which epilogue amplifies its output writes)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "commu-code_amd"))
import torch
from commu_amd import ops
M = 65536
g = torch.Generator(device="cuda").manual_seed(1)
def rnd(*s): return torch.randn(*s, device="cuda", generator=g).bfloat16()
cases = [("plain 512x512 (pipelined)", 512, 512, {}),
         ("resid 512x512 (burst)", 512, 512, {"resid": True}),
         ("plain 1536x512 (pipelined)", 1536, 512, {}),
         ("resid 512x1536 (burst)", 512, 1536, {"resid": True}),
         ("relu 1024x512 (pipelined)", 1024, 512, {"relu": True}),
         ("relumask 1024x512 (burst)", 1024, 512, {"mask": True}),
         ("resid 512x1024 (burst)", 512, 1024, {"resid": True})]
for name, N, K, opt in cases:
    A, W = rnd(M, K), rnd(N, K)
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    aux = rnd(M, N) if opt.get("resid") or opt.get("mask") else None
    torch.cuda.synchronize()
    ops.gemm_nt(A, W, out=out, resid=aux if opt.get("resid") else None, relu=bool(opt.get("relu")),
                relu_mask=aux if opt.get("mask") else None)
    torch.cuda.synchronize()
    print(name, "algorithmic write MB", M * N * 2 / 1e6, "read MB", (M * K + N * K + (M * N if aux is not None else 0)) * 2 / 1e6)
