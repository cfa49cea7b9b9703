"""Does a power-of-two leading dimension cost the eight-phase NT GEMM?  Same product with the A / W rows padded by PAD elements
(row stride K + PAD): if the 8 lines of one staging instruction (8 rows x 128 bytes) fall on one memory channel when the row
stride is a multiple of 4 KB, padding spreads them."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "commu-code_amd"))
import torch
from commu_amd import ops
def t(f, n=20):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
shapes = [(65536, 512, 512), (65536, 1024, 512), (65536, 512, 1024), (65536, 1536, 512), (65536, 512, 1536), (65536, 1536, 4096)]
for (M, N, K) in shapes:
    row = f"NT {M}x{N}x{K}:"
    for pad_a, pad_w, pad_c in [(0, 0, 0), (64, 0, 0), (64, 64, 0), (64, 64, 64), (128, 128, 128), (8, 8, 8)]:
        Ab = torch.randn(M, K + pad_a, device="cuda").bfloat16(); Wb = torch.randn(N, K + pad_w, device="cuda").bfloat16()
        ob = torch.empty(M, N + pad_c, device="cuda", dtype=torch.bfloat16)
        A, W, out = Ab[:, :K], Wb[:, :K], ob[:, :N]
        us = t(lambda: ops.gemm_nt(A, W, out=out))
        row += f"  pad {pad_a},{pad_w},{pad_c}: {us:6.1f} us {2*M*N*K/us/1e6:5.0f} TF |"
        del Ab, Wb, ob
    print(row, flush=True)
