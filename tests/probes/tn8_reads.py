"""Operand bytes of the grouped weight-gradient launch (one launch each, for a FETCH_SIZE pass): algorithmic = every operand
once = 805 MB.  Variants: slice count; the step's problem order (w2, w1 + column sums, o, qkv); the step's operand views
(dY of the qkv problem = a [M, 1536] tensor, X of o = a column slice ...)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "commu-code_amd"))
import torch
from commu_amd import ops
M = 65536
def r(n): return torch.randn(M, n, device="cuda").bfloat16()
shapes = [(1536, 512), (512, 512), (1024, 512), (512, 1024)]
pairs = [(r(N), r(K)) for N, K in shapes]
arr, Mm, offs, total = ops.tn_group(pairs)
print("default slices", ops.tn_group_slices(arr, Mm))
for ns in (4, 8):
    slabs = torch.empty(ns * total, device="cuda")
    torch.cuda.synchronize(); ops.gemm_tn_grouped(arr, Mm, slabs, total, ns); torch.cuda.synchronize()
# the step's order and column sums
step = [pairs[3], pairs[2], pairs[1], pairs[0]]
arr2, Mm, offs2, total2, cs = ops.tn_group(step, colsum=[False, True, False, False])
slabs = torch.empty(4 * total2, device="cuda")
torch.cuda.synchronize(); ops.gemm_tn_grouped(arr2, Mm, slabs, total2, 4); torch.cuda.synchronize()
# the step's order without the column sums
arr3, Mm, offs3, total3 = ops.tn_group(step)
torch.cuda.synchronize(); ops.gemm_tn_grouped(arr3, Mm, slabs, total3, 4); torch.cuda.synchronize()
# the probe's order with column sums on the qkv problem (12 tiles, 6 of them with the extra MFMAs)
arr4, Mm, offs4, total4, cs4 = ops.tn_group(pairs, colsum=[True, False, False, False])
slabs4 = torch.empty(4 * total4, device="cuda")
torch.cuda.synchronize(); ops.gemm_tn_grouped(arr4, Mm, slabs4, total4, 4); torch.cuda.synchronize()
# the same launch right after a kernel that WROTE the operands (as in the step: dirty lines / Infinity-Cache state)
for A, B in step:
    A.mul_(1.0); B.mul_(1.0)
torch.cuda.synchronize(); ops.gemm_tn_grouped(arr2, Mm, slabs, total2, 4); torch.cuda.synchronize()
