"""MX-fp8 GEMM (BASELINE.json configs[4]: "fp8 MFMA GEMMs").  The reference has no fp8 path, so the oracle here is the
OCP microscaling arithmetic itself, emulated on the CPU: block scale 2^(floor(log2 amax) - 8) per 32 k, e4m3 elements
rounded to nearest even (torch.float8_e4m3fn), fp32 accumulation.  Checked:
  * the quantiser bit-exactly (element bytes and scale bytes);
  * the GEMM against the product of the DEQUANTISED operands: only the accumulation order and the bf16 rounding of the
    output differ -- relative error < 6e-3;
  * the stated tolerance against the bf16 path (commu_gemm_nt_bf16 on the un-quantised operands): e4m3 keeps 3 mantissa
    bits -- a rounding error uniform in +-2^-4 of the binade, ~2.7 % RMS per operand, ~4 % for the product of two, and
    the errors of a contraction's terms are independent, so the RESULT carries the same ~4 % of its own RMS:
    RMS difference / RMS value < 5e-2 and max abs difference / max abs value < 1e-1."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rnd(*shape, seed=0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


def mx_quant_ref(x):
    rows, K = x.shape
    xb = x.float().view(rows, K // 32, 32)
    amax = xb.abs().amax(-1)
    eb = torch.where(amax > 0, torch.floor(torch.log2(amax)) + 127, torch.zeros_like(amax))
    sb = (eb - 8).clamp(0, 254)
    inv = torch.pow(2.0, 127 - sb)[..., None]
    q = (xb * inv).clamp(-448, 448).to(torch.float8_e4m3fn)
    deq = q.float() * torch.pow(2.0, sb - 127)[..., None]
    return q.view(torch.uint8).view(rows, K), sb.to(torch.uint8), deq.view(rows, K)


def relerr(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


@pytest.mark.parametrize("rows,K", [(5, 32), (130, 512), (64, 1024)])
def test_quant_mxfp8_bit_exact(rows, K):
    from commu_amd import ops
    x = (rnd(rows, K, seed=1) * torch.exp(rnd(rows, 1, seed=2) * 3)).to(torch.bfloat16)      # rows of very different scale
    x[0, :32] = 0                                                                             # an all-zero block
    q_ref, s_ref, _ = mx_quant_ref(x)
    q, s = ops.quant_mxfp8(x.to(DEV))
    assert torch.equal(s.cpu(), s_ref)
    assert torch.equal(q.cpu(), q_ref)


@pytest.mark.parametrize("M,N,K", [(128, 128, 128), (200, 136, 256), (1024, 1536, 512), (300, 729 + 7, 1024)])
def test_gemm_nt_mxfp8(M, N, K):
    from commu_amd import ops
    a = (rnd(M, K, seed=3) * 0.8).to(torch.bfloat16)
    b = (rnd(N, K, seed=4) * 0.05 * (1 + torch.arange(N)[:, None] % 7)).to(torch.bfloat16)      # asymmetric operands
    bias = rnd(N, seed=5)
    _, _, da = mx_quant_ref(a)
    _, _, db = mx_quant_ref(b)
    ref_mx = da.double() @ db.double().t() + bias.double()
    qa, sa = ops.quant_mxfp8(a.to(DEV))
    qb, sb = ops.quant_mxfp8(b.to(DEV))
    out = ops.gemm_nt_mxfp8(qa, sa, qb, sb, bias=bias.to(DEV))
    assert relerr(out, ref_mx) < 6e-3
    out_relu = ops.gemm_nt_mxfp8(qa, sa, qb, sb, bias=bias.to(DEV), relu=True)
    assert relerr(out_relu, ref_mx.clamp_min(0)) < 6e-3
    # stated tolerance against the bf16 GEMM path on the same (un-quantised) operands
    ref_bf16 = ops.gemm_nt(a.to(DEV), b.to(DEV), bias=bias.to(DEV))
    assert relerr(out, ref_bf16) < 1e-1
    d = (out.float() - ref_bf16.float())
    assert float(d.pow(2).mean().sqrt() / ref_bf16.float().pow(2).mean().sqrt()) < 5e-2


def test_gemm_nt_mxfp8_rejects_bad_shapes():
    from commu_amd import ops
    from commu_amd._lib import CommuHipError
    qa = torch.zeros(16, 96, dtype=torch.uint8, device=DEV)
    sa = torch.zeros(16, 4, dtype=torch.uint8, device=DEV)
    with pytest.raises(CommuHipError):
        ops.gemm_nt_mxfp8(qa, sa, qa, sa)          # K % 128 != 0
