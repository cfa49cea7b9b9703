"""Known byte counts for calibrating FETCH_SIZE / WRITE_SIZE (MI355X_MICROARCH.md, HBM section): a 1 GiB device-to-device
copy (reads 1 GiB, writes 1 GiB) and a 1 GiB fill (writes 1 GiB), four times each, on buffers far larger than the 256 MiB
memory-side cache."""
import torch

n = 1 << 28          # fp32 elements = 1 GiB
a = torch.empty(n, device="cuda", dtype=torch.float32).normal_()
b = torch.empty_like(a)
torch.cuda.synchronize()
for _ in range(4):
    b.copy_(a)
torch.cuda.synchronize()
for _ in range(4):
    b.fill_(1.0)
torch.cuda.synchronize()
print("calibration done")
