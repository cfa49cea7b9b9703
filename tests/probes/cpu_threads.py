import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from oracle import xl_ref as X
s = X.XLShape(6, 8, 512, 1024)
for nt in (8, 16, 32, 64):
    torch.set_num_threads(nt)
    p = X.init_params(s, 1); st = X.adam_init(p)
    g = torch.Generator().manual_seed(0)
    ts = []
    for it in range(3):
        stream = torch.randint(2, 729, (1025, 2), generator=g)
        t0 = time.time()
        X.train_step(p, st, s, stream[:-1], stream[1:], torch.zeros(2, dtype=torch.bool), [None], batch_chunk=1, mem_len=0, same_length=False, lr_now=1e-4, clip=1.0)
        ts.append(time.time() - t0)
    print(nt, "threads", ["%.2f" % t for t in ts], flush=True)
