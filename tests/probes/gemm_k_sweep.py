"""Isolated gemm_nt8: time against K (fixed per-tile cost vs per-K-tile cost) and the ablations of one shape."""
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


for N in (512, 1024, 1536):
    for K in (512, 1024, 2048):
        x = torch.randn(M, K, device=dev).bfloat16()
        w = torch.randn(N, K, device=dev).bfloat16()
        y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        t = timeit(lambda: ops.gemm_nt(x, w, out=y))
        tiles = (M // 256) * (N // 256)
        print(f"N {N:5d} K {K:5d}  {t:7.1f} us  {2.0 * M * N * K / t / 1e6:6.0f} TFLOP/s   rounds {tiles / 256:.0f}  per tile {t / (tiles / 256):6.2f} us")
x = torch.randn(M, 512, device=dev).bfloat16()
w = torch.randn(1024, 512, device=dev).bfloat16()
y = torch.empty(M, 1024, device=dev, dtype=torch.bfloat16)
y2 = torch.empty(M, 1024, device=dev, dtype=torch.bfloat16)
b = torch.randn(1024, device=dev)
r = torch.randn(M, 1024, device=dev).bfloat16()
if os.environ.get("SWEEP_ABL"):          # (needs a library built with the COMMU_GEMM8_ABL switch, removed from the product)
    for abl in ("0", "1", "2", "3"):
        os.environ["COMMU_GEMM8_ABL"] = abl
        t = timeit(lambda: ops.gemm_nt(x, w, out=y))
        print(f"(65536,1024,512) plain  ABL {abl}: {t:7.1f} us")
    os.environ["COMMU_GEMM8_ABL"] = "0"
cases = {"plain": {}, "bias+relu+dropout": dict(bias=b, relu=True, drop_p=0.1, drop_seed=3), "resid": dict(resid=r),
         "bias+dropout+resid": dict(bias=b, drop_p=0.1, drop_seed=3, resid=r)}
for name, kw in cases.items():
    os.environ["COMMU_GEMM8_NOPIPE"] = "1"
    t0 = timeit(lambda: ops.gemm_nt(x, w, out=y, **kw))
    os.environ.pop("COMMU_GEMM8_NOPIPE", None)
    t1 = timeit(lambda: ops.gemm_nt(x, w, out=y2, **kw))
    print(f"(65536,1024,512) {name:20s} burst {t0:7.1f} us   pipelined {t1:7.1f} us   equal {bool(torch.equal(y, y2))}"
          f"  max|diff| {float((y.float() - y2.float()).abs().max()):.4f} of max {float(y.float().abs().max()):.1f}")
for N, K in ((512, 512), (512, 1024), (1536, 512), (512, 1536)):
    x = torch.randn(M, K, device=dev).bfloat16()
    w = torch.randn(N, K, device=dev).bfloat16()
    y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    y2 = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    os.environ["COMMU_GEMM8_NOPIPE"] = "1"
    t0 = timeit(lambda: ops.gemm_nt(x, w, out=y))
    os.environ.pop("COMMU_GEMM8_NOPIPE", None)
    t1 = timeit(lambda: ops.gemm_nt(x, w, out=y2))
    print(f"(65536,{N},{K}) plain   burst {t0:7.1f} us   pipelined {t1:7.1f} us   equal {bool(torch.equal(y, y2))}")
# ReLU backward: bf16 mask against one bit per element (shape of dhid = dz . W2: M x 1024, K = 512)
os.environ.pop("COMMU_GEMM8_NOPIPE", None)
x = torch.randn(M, 512, device=dev).bfloat16()
w = torch.randn(1024, 512, device=dev).bfloat16()
y = torch.empty(M, 1024, device=dev, dtype=torch.bfloat16)
y2 = torch.empty(M, 1024, device=dev, dtype=torch.bfloat16)
b = torch.randn(1024, device=dev)
bits = torch.empty(M * 1024 // 32, device=dev, dtype=torch.int32)
kw = dict(bias=b, relu=True, drop_p=0.1, drop_seed=3)
t0 = timeit(lambda: ops.gemm_nt(x, w, out=y, **kw))
t1 = timeit(lambda: ops.gemm_nt(x, w, out=y2, sign_bits_out=bits, **kw))
print(f"forward bias+relu+dropout {t0:7.1f} us   + sign bits out {t1:7.1f} us   equal {bool(torch.equal(y, y2))}")
hid = y.clone()
t0 = timeit(lambda: ops.gemm_nt(x, w, out=y, relu_mask=hid, mask_scale=1.1))
t1 = timeit(lambda: ops.gemm_nt(x, w, out=y2, relu_bits=bits, mask_scale=1.1))
print(f"backward relu_mask (bf16) {t0:7.1f} us   relu_bits {t1:7.1f} us   equal {bool(torch.equal(y, y2))}")
