"""Attention backward of one layer at the bench shape: the whole batch at once against batch slices that re-use ONE
(smaller) dS-by-distance / P scratch -- does a scratch that fits the 256-MB memory-side cache save HBM time?"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "commu-code_amd"))
from commu_amd import ops
T, M, H, DH, BT = 1024, 0, 8, 64, 64
HD, K = H * DH, T + M
dev = "cuda"
DROP = 0.1


def make(B, seed):
    g = torch.Generator().manual_seed(seed)
    qkv = (torch.randn(K * B, 3 * HD, generator=g) * 0.7).to(torch.bfloat16).to(dev)
    rd = (torch.randn(K, HD, generator=g) * 0.7).to(torch.bfloat16).to(dev)
    u = (torch.randn(HD, generator=g) * 0.3).to(dev)
    vb = (torch.randn(HD, generator=g) * 0.3).to(dev)
    dout = torch.randn(T * B, HD, generator=g).to(torch.bfloat16).to(dev)
    q, k, v = qkv[M * B:, :HD], qkv[:, HD:2 * HD], qkv[:, 2 * HD:]
    out, lse, qs = ops.relattn_fwd(q, k, v, rd, u, vb, None, T, M, B, H, DH, False, M, save_q=True, drop_p=DROP, drop_seed=1234)
    dqkv = torch.zeros_like(qkv)
    return dict(q=q, k=k, v=v, rd=rd, u=u, vb=vb, out=out, dout=dout, lse=lse, qs=qs, dqkv=dqkv,
                drd=torch.zeros(K, HD, device=dev), du=torch.zeros(HD, device=dev), dvb=torch.zeros(HD, device=dev), B=B)


def bwd(s, scratch):
    B = s["B"]
    ops.relattn_bwd(s["q"], s["k"], s["v"], s["rd"], s["u"], s["vb"], None, T, M, B, H, DH, False, M, s["out"], s["dout"],
                    s["lse"], s["qs"], s["dqkv"][M * B:, :HD], s["dqkv"][:, HD:2 * HD], s["dqkv"][:, 2 * HD:], s["drd"],
                    s["du"], s["dvb"], drop_p=DROP, drop_seed=1234, scratch=scratch)


for Bs in (64, 32, 16, 8):
    sets = [make(Bs, i) for i in range(BT // Bs)]
    scratch = {}
    for s in sets:
        bwd(s, scratch)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        for s in sets:
            bwd(s, scratch)
    e1.record()
    torch.cuda.synchronize()
    mb = sum(v.numel() * v.element_size() for v in scratch.values() if torch.is_tensor(v)) / 2**20
    print(f"slices of {Bs:2d} sequences x {BT // Bs}: {e0.elapsed_time(e1) / 5 * 1e3:8.1f} us per layer backward   scratch {mb:.0f} MiB")
    del sets, scratch
    torch.cuda.empty_cache()
