"""Does the Infinity Cache (256 MB, memory side) keep freshly WRITTEN data for the next kernel?  Read rate of a buffer right after
a kernel wrote it, against the same read after 2 GB of other traffic, for several sizes (our own streaming kernels:
commu layernorm is not needed -- torch's sum / fill are bandwidth kernels)."""
import time, torch
dev = "cuda"
flush = torch.empty(1 << 29, device=dev, dtype=torch.float32)          # 2 GiB
def t_read(a, warm):
    ts = []
    for _ in range(5):
        if warm:
            a.fill_(1.0)
        else:
            a.fill_(1.0); flush.fill_(0.0)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); s = a.sum(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e-3)
    return sorted(ts)[len(ts) // 2]
for mb in (16, 32, 64, 128, 192, 256, 384, 512, 1024):
    a = torch.empty(mb * (1 << 20) // 4, device=dev, dtype=torch.float32)
    tw, tc = t_read(a, True), t_read(a, False)
    print(f"{mb:5d} MB: read right after the write {mb / 1024 / tw:7.2f} GB/ms ({tw * 1e6:7.1f} us) | after a 2-GiB flush {mb / 1024 / tc:7.2f} GB/ms ({tc * 1e6:7.1f} us)", flush=True)
