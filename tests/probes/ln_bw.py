"""Isolated bandwidth of the LayerNorm backward / reduction kernels at the bench shape ([65536, 512] bf16)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "commu-code_amd"))
from commu_amd import ops

dev = torch.device("cuda:0")
rows, D = 65536, 512
torch.manual_seed(0)
dy = torch.randn(rows, D, device=dev).bfloat16()
z = torch.randn(rows, D, device=dev).bfloat16()
mu = z.float().mean(1).contiguous()
rs = (z.float().var(1, unbiased=False) + 1e-5).rsqrt().contiguous()
gamma = torch.ones(D, device=dev)
dzm = torch.empty_like(z)
dg, db, dbias = (torch.zeros(D, device=dev) for _ in range(3))
big = torch.randn(rows, 1024, device=dev).bfloat16()
ob = torch.zeros(1024, device=dev)


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


dz, part = ops.layernorm_bwd(dy, z, mu, rs, gamma, dz_masked=dzm, drop_p=0.1, drop_seed=7)
t = timeit(lambda: ops.layernorm_bwd(dy, z, mu, rs, gamma, dz=dz, part=part, dz_masked=dzm, drop_p=0.1, drop_seed=7))
print(f"layernorm_bwd (+masked copy) {t:7.1f} us  {4 * rows * D * 2 / t / 1e6:6.2f} TB/s   partials {tuple(part.shape)}")
t = timeit(lambda: ops.layernorm_bwd(dy, z, mu, rs, gamma, dz=dz, part=part))
print(f"layernorm_bwd (no dropout)   {t:7.1f} us  {3 * rows * D * 2 / t / 1e6:6.2f} TB/s")
t = timeit(lambda: ops.layernorm_bwd_reduce(part, dg, db, dbias))
print(f"layernorm_bwd_reduce         {t:7.1f} us")
t = timeit(lambda: ops.colsum(big, ob))
print(f"colsum [65536,1024] bf16     {t:7.1f} us  {rows * 1024 * 2 / t / 1e6:6.2f} TB/s")
# embedding backward at the bench size (65 536 tokens, V = 729, D = 512), dropout on
tok = torch.randint(0, 729, (rows,), device=dev)
dE = torch.zeros(729, D, device=dev)
order = ops.token_order(tok, 729)
t = timeit(lambda: ops.embed_bwd(tok, dy, dE, accumulate=True, drop_p=0.1, drop_seed=5, order=order))
print(f"embed_bwd (sorted + reduce)  {t:7.1f} us")
t = timeit(lambda: ops.token_order(tok, 729))
print(f"token_order (torch)          {t:7.1f} us")
t = timeit(lambda: ops.embed_bwd(tok, dy, dE, accumulate=True, drop_p=0.1, drop_seed=5))
print(f"embed_bwd (row scan)         {t:7.1f} us")
