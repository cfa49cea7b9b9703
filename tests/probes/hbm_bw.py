"""HBM bandwidth of this part for pure writes, pure reads and copies (1 GiB buffers, torch kernels)."""
import time, torch
dev = "cuda"
n = 1 << 29          # bf16 elements = 1 GiB
a = torch.empty(n, device=dev, dtype=torch.bfloat16); b = torch.empty(n, device=dev, dtype=torch.bfloat16)
def t(f, reps=10):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
tw = t(lambda: a.zero_()); tr = t(lambda: a.sum()); tc = t(lambda: b.copy_(a))
gib = n * 2 / 1e9
print(f"write {gib/tw/1e3:.2f} TB/s ({tw*1e6:.0f} us / GiB) | read {gib/tr/1e3:.2f} TB/s | copy {2*gib/tc/1e3:.2f} TB/s (read+write)")
