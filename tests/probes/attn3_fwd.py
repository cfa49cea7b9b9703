"""Forward attention at the bench shape: generation 2 (16x16 layout) against generation 3 (32x32 / transposed scores),
with and without attention dropout; also compares the two kernels' outputs (no dropout).  One subprocess per setting
(the generation switch is read once per process)."""
import os, subprocess, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "commu-code_amd"))
T, M, H, DH, B = 1024, int(os.environ.get("PROBE_M", "0")), 8, 64, 64
HD, K = H * DH, T + M


def run(drop):
    from commu_amd import ops
    ops.attn_fwd_generation(int(os.environ.get("COMMU_ATTN_FWD_GEN", "0")))
    dev = "cuda"
    g = torch.Generator().manual_seed(1)
    qkv = (torch.randn(K * B, 3 * HD, generator=g) * 0.7).to(torch.bfloat16).to(dev)
    rd = (torch.randn(K, HD, generator=g) * 0.7).to(torch.bfloat16).to(dev)
    u = (torch.randn(HD, generator=g) * 0.3).to(dev)
    vb = (torch.randn(HD, generator=g) * 0.3).to(dev)
    q, k, v = qkv[M * B:, :HD], qkv[:, HD:2 * HD], qkv[:, 2 * HD:]
    out = lse = None
    for _ in range(3):
        out, lse, qs = ops.relattn_fwd(q, k, v, rd, u, vb, None, T, M, B, H, DH, False, M, save_q=True, drop_p=drop, drop_seed=1234)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.relattn_fwd(q, k, v, rd, u, vb, None, T, M, B, H, DH, False, M, save_q=True, drop_p=drop, drop_seed=1234)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    gf = 6.0 * (M + (T + 1) / 2) * HD * T * B / 1e9
    print(f"gen {os.environ.get('COMMU_ATTN_FWD_GEN')} drop {drop}: {us:8.1f} us  {gf / us * 1e3:6.1f} TFLOP/s"
          f"  out sum {float(out.float().abs().sum()):.6e} lse sum {float(lse.sum()):.6e}", flush=True)
    if drop == 0.0:
        torch.save((out.cpu(), lse.cpu()), f"/tmp/attn3_out_gen{os.environ.get('COMMU_ATTN_FWD_GEN')}.pt")


if __name__ == "__main__":
    if len(sys.argv) > 1:
        run(float(sys.argv[1]))
    else:
        for gen in ("2", "3"):
            for drop in ("0.0", "0.1"):
                env = dict(os.environ, COMMU_ATTN_FWD_GEN=gen)
                subprocess.run([sys.executable, __file__, drop], env=env, check=False)
        a, b = torch.load("/tmp/attn3_out_gen2.pt"), torch.load("/tmp/attn3_out_gen3.pt")
        print("gen3 vs gen2: out max diff", float((a[0].float() - b[0].float()).abs().max()),
              "lse max diff", float((a[1] - b[1]).abs().max()))
