import sys, torch
sys.path.insert(0, "commu-code_amd")
from commu_amd import ops
dev = "cuda"
ONE = 0x38
B = torch.full((16, 128), ONE, dtype=torch.uint8, device=dev)
SB = torch.full((16, 4), 127, dtype=torch.uint8, device=dev)
for lo in range(0, 128, 16):
    A = torch.zeros(16, 128, dtype=torch.uint8, device=dev)
    A[:, lo:lo + 16] = ONE
    SA = torch.tensor([[127, 128, 129, 130]] * 16, dtype=torch.uint8, device=dev)
    out = ops.gemm_nt_mxfp8(A, SA, B, SB)
    print("A ones in bytes", lo, "..", lo + 15, "-> C[0,0] =", float(out[0, 0]), " => scale applied", float(out[0, 0]) / 16)
# now B side scale probe
A = torch.full((16, 128), ONE, dtype=torch.uint8, device=dev)
SA = torch.full((16, 4), 127, dtype=torch.uint8, device=dev)
for lo in range(0, 128, 16):
    Bm = torch.zeros(16, 128, dtype=torch.uint8, device=dev)
    Bm[:, lo:lo + 16] = ONE
    SBm = torch.tensor([[127, 128, 129, 130]] * 16, dtype=torch.uint8, device=dev)
    out = ops.gemm_nt_mxfp8(A, SA, Bm, SBm)
    print("B ones in bytes", lo, "->", float(out[0, 0]) / 16)
