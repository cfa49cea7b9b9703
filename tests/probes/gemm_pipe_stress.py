"""Stress of the pipelined epilogue's counted-wait assumption (vmcnt retires loads and stores in order): pipelined against
burst results, bit for bit, over many launches while a second stream keeps the memory system busy."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "commu-code_amd"))
from commu_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
side = torch.cuda.Stream()
big_a = torch.randn(64 * 1024 * 1024, device=dev)
big_b = torch.empty_like(big_a)
bad = 0
n = 0
for it in range(60):
    M = 256 * int(torch.randint(64, 257, (1,)))
    N = 256 * int(torch.randint(1, 7, (1,)))
    K = 64 * int(torch.randint(2, 17, (1,)))
    x = torch.randn(M, K, device=dev).bfloat16()
    w = torch.randn(N, K, device=dev).bfloat16()
    b = torch.randn(N, device=dev)
    kw = [dict(), dict(bias=b, relu=True), dict(bias=b, relu=True, drop_p=0.1, drop_seed=it)][it % 3]
    os.environ["COMMU_GEMM8_NOPIPE"] = "1"
    ref = ops.gemm_nt(x, w, **kw)
    os.environ.pop("COMMU_GEMM8_NOPIPE")
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        for _ in range(4):
            big_b.copy_(big_a)          # HBM traffic beside the GEMMs
    outs = [ops.gemm_nt(x, w, **kw) for _ in range(6)]
    torch.cuda.synchronize()
    for o in outs:
        n += 1
        if not torch.equal(o, ref):
            bad += 1
            print("MISMATCH", M, N, K, it % 3, int((o != ref).sum()))
print(f"{n} pipelined launches compared, {bad} mismatches")
