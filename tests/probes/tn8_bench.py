"""Grouped eight-phase TN (weight gradients of one layer in one launch) vs the per-GEMM path, bench shape."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "commu-code_amd"))
import torch
from commu_amd import ops
def t(f, n=10):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
M = 65536
shapes = [(1536, 512), (512, 512), (1024, 512), (512, 1024)]
pairs = [(torch.randn(M, N, device="cuda").bfloat16(), torch.randn(M, K, device="cuda").bfloat16()) for N, K in shapes]
outs = [torch.zeros(N, K, device="cuda") for N, K in shapes]
fl = sum(2.0 * M * N * K for N, K in shapes)
arr, Mm, offs, total = ops.tn_group(pairs)
for ns in (ops.tn_group_slices(arr, Mm), 4, 16):
    slabs = torch.empty(ns * total, device="cuda")
    def grouped(reduce=True):
        ops.gemm_tn_grouped(arr, Mm, slabs, total, ns)
        if reduce:
            for (N, K), off, o in zip(shapes, offs, outs):
                ops.reduce_slabs(o, slabs[off:], N * K, ns, total, True)
    us0 = t(lambda: grouped(False)); us1 = t(grouped)
    print(f"grouped layer, {ns} slices: kernel {us0:7.1f} us {fl/us0/1e6:6.0f} TF | with reduces {us1:7.1f} us {fl/us1/1e6:6.0f} TF", flush=True)
def separate():
    for (A, B), o in zip(pairs, outs):
        ops.gemm_tn(A, B, o, accumulate=True)
us = t(separate)
print(f"per-GEMM path (old kernels + reduce): {us:7.1f} us {fl/us/1e6:6.0f} TF")
# single problems through the grouped kernel
for (A, B), (N, K) in zip(pairs, shapes):
    arr1, _, _, tot1 = ops.tn_group([(A, B)])
    ns = ops.tn_group_slices(arr1, M)
    slabs = torch.empty(ns * tot1, device="cuda")
    us = t(lambda: ops.gemm_tn_grouped(arr1, M, slabs, tot1, ns))
    print(f"TN {N}x{K}x{M} alone, {ns} slices: {us:7.1f} us {2.0*M*N*K/us/1e6:6.0f} TF")
