"""NT GEMMs of the training step with their real epilogues, isolated, against torch.matmul (vendor BLAS) on the same shapes."""
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
cases = [  # name, N, K, kwargs builder
    ("qkv fwd plain", 1536, 512, {}),
    ("o fwd resid+drop", 512, 512, dict(resid=True, drop=True)),
    ("ff1 fwd bias+relu+drop", 1024, 512, dict(bias=True, relu=True, drop=True)),
    ("ff2 fwd bias+resid+drop", 512, 1024, dict(bias=True, resid=True, drop=True)),
    ("dhid = dz2.W2 relumask", 1024, 512, dict(mask=True)),
    ("da = dhid.W1 resid", 512, 1024, dict(resid=True)),
    ("dvec = dz1.Wo plain", 512, 512, {}),
    ("dy = dqkv.Wqkv resid", 512, 1536, dict(resid=True)),
    ("dy = dlogits.E drop", 512, 768, dict(drop=True)),
    ("logits f32 bias", 729, 512, dict(bias=True, f32=True)),
]
for name, N, K, kw in cases:
    A = torch.randn(M, K, device="cuda").bfloat16(); W = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    ld = 768 if N == 729 else N
    out = torch.empty(M, ld, device="cuda", dtype=torch.float32 if kw.get("f32") else torch.bfloat16)[:, :N]
    bias = torch.randn(N, device="cuda") if kw.get("bias") else None
    resid = torch.randn(M, N, device="cuda").bfloat16() if kw.get("resid") else None
    mask = torch.randn(M, N, device="cuda").bfloat16() if kw.get("mask") else None
    f = lambda: ops.gemm_nt(A, W, out=out, bias=bias, resid=resid, relu=bool(kw.get("relu")), relu_mask=mask,
                            drop_p=0.1 if kw.get("drop") else 0.0, drop_seed=123, mask_scale=1.11)
    us = t(f)
    ref = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    Wt = W.t().contiguous()
    us_blas = t(lambda: torch.matmul(A, W.t(), out=ref))
    fl = 2 * M * N * K
    print(f"{name:28s} N={N:5d} K={K:5d}: {us:7.1f} us {fl/us/1e6:5.0f} TF | torch.matmul plain {us_blas:7.1f} us {fl/us_blas/1e6:5.0f} TF", flush=True)
    del A, W, out, bias, resid, mask, ref
