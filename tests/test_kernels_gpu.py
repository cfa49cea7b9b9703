"""GPU parity tests of the individual HIP kernels, called through the C ABI
(commu_amd.ops -> commu_amd._lib -> libcommu_hip.so), against the CPU oracle's math evaluated
in fp32/fp64 on the same bf16-rounded inputs.

Tolerances (stated per check): results stored as bf16 carry one rounding (2^-9 relative), fp32
results differ from the oracle only by accumulation order -> `relerr` = max|a-b| / max|b|.
"""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import xl_ref as X  # noqa: E402

DEV = "cuda"
BF16_TOL = 1.2e-2      # one or two bf16 roundings of O(1)-relative values
F32_TOL = 2e-3         # fp32 outputs of bf16-input GEMMs / reductions (order only)


def ops():
    from commu_amd import ops as o
    return o


def relerr(a, b):
    a = a.detach().float().cpu()
    b = b.detach().float().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def bf(x):
    return x.to(torch.bfloat16)


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (300, 200, 64), (257, 729, 128), (64, 1536, 512), (1, 96, 64)])
def test_gemm_nt_plain(M, N, K):
    o = ops()
    A, B = bf(rnd(M, K, seed=1)), bf(rnd(N, K, seed=2))
    ref = A.float() @ B.float().t()
    ld = (N + 7) // 8 * 8                           # output leading dimension must be a multiple of 4
    out = o.gemm_nt(A.to(DEV), B.to(DEV), out=torch.empty(M, ld, device=DEV)[:, :N])
    assert relerr(out, ref) < F32_TOL
    out = o.gemm_nt(A.to(DEV), B.to(DEV), out=torch.empty(M, ld, device=DEV, dtype=torch.bfloat16)[:, :N])
    assert out.dtype == torch.bfloat16 and relerr(out, ref) < BF16_TOL


def test_gemm_nt_asymmetric_identity():
    """A = I with an asymmetric B catches a transposed C write (guide rule 16)."""
    o = ops()
    K = 64
    A = bf(torch.eye(K))
    B = bf(torch.arange(96 * K, dtype=torch.float32).reshape(96, K) % 251)
    out = o.gemm_nt(A.to(DEV), B.to(DEV), out_f32=True)
    assert torch.equal(out.cpu(), B.float().t())


def test_gemm_nt_epilogues():
    o = ops()
    M, N, K = 200, 264, 96
    A, B = bf(rnd(M, K, seed=3)), bf(rnd(N, K, seed=4))
    bias, resid, act = rnd(N, seed=5), bf(rnd(M, N, seed=6)), bf(rnd(M, N, seed=7))
    ref = A.float() @ B.float().t()
    out = o.gemm_nt(A.to(DEV), B.to(DEV), bias=bias.to(DEV), relu=True, out_f32=True)
    assert relerr(out, torch.relu(ref + bias)) < F32_TOL
    out = o.gemm_nt(A.to(DEV), B.to(DEV), bias=bias.to(DEV), resid=resid.to(DEV), out_f32=True)
    assert relerr(out, ref + bias + resid.float()) < F32_TOL
    out = o.gemm_nt(A.to(DEV), B.to(DEV), relu_mask=act.to(DEV), out_f32=True)
    assert relerr(out, ref * (act.float() > 0)) < F32_TOL
    # strided views (column slices of wider buffers)
    wide = torch.zeros(M, 400, dtype=torch.bfloat16, device=DEV)
    o.gemm_nt(A.to(DEV), B.to(DEV), out=wide[:, 100:100 + N])
    assert relerr(wide[:, 100:100 + N], ref) < BF16_TOL and float(wide[:, :100].abs().max()) == 0


@pytest.mark.parametrize("M,N,K,D", [(64, 1536, 512, 512), (5, 729, 512, 500), (33, 1024, 1024, 1024), (64, 96, 128, 100)])
def test_gemm_nt_layernorm_fused_decode_form(M, N, K, D):
    """commu_gemm_nt_ln_bf16 (LayerNorm inside the decode-step Linear) against layernorm_fwd + gemm_nt: the stored
    LN rows and the product, incl. zero-padded model widths (D < K) and the bias / relu / residual epilogue."""
    o = ops()
    o.FUSE_DECODE_LN = True          # (off by default: slower than the separate LayerNorm kernel in the decode step)
    z = bf(rnd(M, K, seed=50) * 1.7 + 0.3)
    z[:, D:] = 0
    gamma, beta = rnd(D, seed=51) * 0.2 + 1.0, rnd(D, seed=52) * 0.1
    W, bias, resid = bf(rnd(N, K, seed=53) * 0.1), rnd(N, seed=54), bf(rnd(M, N, seed=55))
    zd, Wd = z.to(DEV), W.to(DEV)
    a_ref, _, _ = o.layernorm_fwd(zd, gamma.to(DEV), beta.to(DEV))
    ld = (N + 7) // 8 * 8
    for kw in (dict(), dict(bias=bias.to(DEV), relu=True), dict(bias=bias.to(DEV), resid=resid.to(DEV))):
        ref = o.gemm_nt(a_ref, Wd, out=torch.empty(M, ld, device=DEV)[:, :N], **kw)
        a_out = torch.full((M, K), float("nan"), device=DEV, dtype=torch.bfloat16)
        out = o.gemm_nt_ln(zd, gamma.to(DEV), beta.to(DEV), Wd, out=torch.empty(M, ld, device=DEV)[:, :N], a_out=a_out, **kw)
        # (row statistics are reduced in a different order: LN rows may differ by one bf16 ulp)
        assert relerr(a_out, a_ref) < 8e-3 and float(a_out[:, D:].abs().max() if D < K else 0) == 0
        assert relerr(out, ref) < 4e-3, kw.keys()
    o.FUSE_DECODE_LN = False
    xf = z.float()[:, :D]
    want = (xf - xf.mean(1, keepdim=True)) / torch.sqrt(xf.var(1, unbiased=False, keepdim=True) + 1e-5) * gamma + beta
    assert relerr(a_out[:, :D], want) < BF16_TOL


# the 256 x 256 x 64 eight-phase kernel (gemm8.hip): interior tiles, M / N tails, odd K-tile counts, a persistent
# grid smaller than the tile count (COMMU_GEMM8_GRID), every epilogue
@pytest.mark.parametrize("grid", [0, 3])
@pytest.mark.parametrize("M,N,K", [(2048, 512, 512), (2300, 729, 512), (1041, 1536, 128), (1024, 264, 192),
                                   (4096, 1024, 1024)])
def test_gemm_nt_eight_phase(M, N, K, grid, monkeypatch):
    o = ops()
    monkeypatch.setenv("COMMU_GEMM8_ALWAYS", "1")          # (the dispatcher only takes it from 256 output tiles on)
    if grid:
        monkeypatch.setenv("COMMU_GEMM8_GRID", str(grid))
    A, B = bf(rnd(M, K, seed=21)), bf(rnd(N, K, seed=22))
    ref = A.float() @ B.float().t()
    ld = (N + 7) // 8 * 8
    Ad, Bd = A.to(DEV), B.to(DEV)
    out = o.gemm_nt(Ad, Bd, out=torch.full((M, ld), float("nan"), device=DEV)[:, :N])
    assert relerr(out, ref) < F32_TOL
    out = o.gemm_nt(Ad, Bd, out=torch.full((M, ld), float("nan"), device=DEV, dtype=torch.bfloat16)[:, :N])
    assert relerr(out, ref) < BF16_TOL
    def buf(dtype=torch.float32):          # NaN-filled, leading dimension padded to a multiple of 8
        return torch.full((M, ld), float("nan"), device=DEV, dtype=dtype)[:, :N]

    def padded(x):                         # same for the auxiliary operands (strided views)
        t = torch.zeros(M, ld, dtype=x.dtype, device=DEV)
        t[:, :N] = x.to(DEV)
        return t[:, :N]
    bias, resid, act = rnd(N, seed=23), bf(rnd(M, N, seed=24)), bf(rnd(M, N, seed=25))
    out = o.gemm_nt(Ad, Bd, out=buf(), bias=bias.to(DEV), relu=True)
    assert relerr(out, torch.relu(ref + bias)) < F32_TOL
    out = o.gemm_nt(Ad, Bd, out=buf(), bias=bias.to(DEV), resid=padded(resid))
    assert relerr(out, ref + bias + resid.float()) < F32_TOL
    out = o.gemm_nt(Ad, Bd, out=buf(torch.bfloat16), relu_mask=padded(act), mask_scale=1.25)
    assert relerr(out, ref * (act.float() > 0) * 1.25) < BF16_TOL
    # dropout epilogue: the kernel's counter-based mask, re-created on the host
    seed, p = 12345, 0.1
    keep = o.dropout_keep_mask(seed, M * N, p).view(M, N)
    out = o.gemm_nt(Ad, Bd, out=buf(), resid=padded(resid), drop_p=p, drop_seed=seed)
    assert relerr(out, ref * keep / (1 - p) + resid.float()) < F32_TOL


@pytest.mark.parametrize("grid", [0, 2, 5])
@pytest.mark.parametrize("M,N,K", [(2048, 512, 128), (2300, 729, 192), (1041, 1536, 512)])
def test_gemm_nt_eight_phase_pipelined_epilogue(M, N, K, grid, monkeypatch):
    """bf16 output without an auxiliary operand: interior waves write a finished tile during the next tile's first K-tile
    (gemm8.hip g8_drain).  Bit-identical to the burst epilogue (COMMU_GEMM8_NOPIPE) for plain, bias+relu and
    bias+relu+dropout, with several tiles per workgroup (grid 2, 5), edge tiles in the mix, K-tiles 2, 3, 8."""
    o = ops()
    monkeypatch.setenv("COMMU_GEMM8_ALWAYS", "1")
    if grid:
        monkeypatch.setenv("COMMU_GEMM8_GRID", str(grid))
    A, B, bias = bf(rnd(M, K, seed=31)).to(DEV), bf(rnd(N, K, seed=32)).to(DEV), rnd(N, seed=33).to(DEV)
    ld = (N + 7) // 8 * 8
    ref = A.float() @ B.float().t()

    def run(**kw):
        out = torch.full((M, ld), float("nan"), device=DEV, dtype=torch.bfloat16)[:, :N]
        return o.gemm_nt(A, B, out=out, **kw)
    cases = [dict(), dict(bias=bias, relu=True), dict(bias=bias, relu=True, drop_p=0.1, drop_seed=77)]
    pipe = [run(**kw) for kw in cases]
    monkeypatch.setenv("COMMU_GEMM8_NOPIPE", "1")
    burst = [run(**kw) for kw in cases]
    for a_, b_ in zip(pipe, burst):
        assert torch.equal(a_, b_)
    assert relerr(pipe[0], ref) < BF16_TOL
    assert relerr(pipe[1], torch.relu(ref + bias)) < BF16_TOL
    keep = o.dropout_keep_mask(77, M * N, 0.1).view(M, N).to(ref.device)
    assert relerr(pipe[2], torch.relu(ref + bias) * keep / 0.9) < BF16_TOL


@pytest.mark.parametrize("grid", [0, 3])
@pytest.mark.parametrize("M,N,K,K2", [(2048, 512, 128, 256), (1024, 1024, 512, 192)])
def test_gemm_nt_relu_sign_bits(M, N, K, K2, grid, monkeypatch):
    """ReLU backward from one bit per element: the forward GEMM's epilogue writes (out > 0) into a word buffer, the
    backward GEMM (same M x N output) reads it instead of the bf16 activations.  Bit-identical to the relu_mask path."""
    o = ops()
    monkeypatch.setenv("COMMU_GEMM8_ALWAYS", "1")
    if grid:
        monkeypatch.setenv("COMMU_GEMM8_GRID", str(grid))
    assert o.signbits_words(2300, 729, 128) == 0 and o.signbits_words(M, N, K) == M * N // 32
    A, W1, b1 = bf(rnd(M, K, seed=41)).to(DEV), bf(rnd(N, K, seed=42)).to(DEV), rnd(N, seed=43).to(DEV)
    G, W2t = bf(rnd(M, K2, seed=44)).to(DEV), bf(rnd(N, K2, seed=45)).to(DEV)
    kw = dict(bias=b1, relu=True, drop_p=0.1, drop_seed=5)
    hid = o.gemm_nt(A, W1, **kw)
    bits = torch.full((M * N // 32,), -1, device=DEV, dtype=torch.int32)
    hid2 = o.gemm_nt(A, W1, sign_bits_out=bits, **kw)
    assert torch.equal(hid, hid2)
    assert int((hid > 0).sum()) == sum(bin(int(x) & 0xFFFFFFFF).count("1") for x in bits[:4096].tolist()) \
        + int(sum(bin(int(x) & 0xFFFFFFFF).count("1") for x in bits[4096:].tolist()))
    ref = o.gemm_nt(G, W2t, relu_mask=hid, mask_scale=1.0 / 0.9)
    got = o.gemm_nt(G, W2t, relu_bits=bits, mask_scale=1.0 / 0.9)
    assert torch.equal(ref, got)
    full = (G.float() @ W2t.float().t()) * (hid.float() > 0) / 0.9
    assert relerr(got, full.cpu()) < BF16_TOL


def test_gemm_nt_eight_phase_asymmetric_identity(monkeypatch):
    """A = [I; 0...] pattern with an asymmetric B: catches transposed / permuted C writes of the 8-phase kernel."""
    o = ops()
    monkeypatch.setenv("COMMU_GEMM8_ALWAYS", "1")
    M, N, K = 1024, 512, 128
    A = torch.zeros(M, K)
    for m in range(M):
        A[m, (7 * m + 3) % K] = 1.0
    B = (torch.arange(N * K, dtype=torch.float32).reshape(N, K) * 37 % 251)
    out = o.gemm_nt(bf(A).to(DEV), bf(B).to(DEV), out_f32=True)
    assert torch.equal(out.cpu(), bf(A).float() @ bf(B).float().t())


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("M,N,K", [(64, 128, 128), (1000, 256, 192), (333, 136, 64), (4096, 768, 64), (50, 8, 8)])
def test_gemm_tn(M, N, K, mode):
    o = ops()
    A, B = bf(rnd(M, N, seed=8)), bf(rnd(M, K, seed=9))
    ref = A.float().t() @ B.float()
    out = torch.zeros(N, K, device=DEV)
    o.gemm_tn(A.to(DEV), B.to(DEV), out, mode=mode)
    assert relerr(out, ref) < F32_TOL
    o.gemm_tn(A.to(DEV), B.to(DEV), out, accumulate=True, mode=mode)
    assert relerr(out, 2 * ref) < F32_TOL


@pytest.mark.parametrize("M", [4096, 5000])
def test_gemm_tn_grouped_eight_phase(M):
    """All weight gradients of a layer in ONE launch of the eight-phase TN kernel (gemm8.hip): mixed shapes, a
    partial output tile (N = 328), a strided operand view, token-slice tails (M = 5000)."""
    o = ops()
    wide = bf(rnd(M, 1536, seed=30))
    shapes = [(1536, 512), (512, 512), (1024, 512), (512, 1024), (328, 136)]
    pairs, refs = [], []
    for i, (N, K) in enumerate(shapes):
        A = wide[:, :N] if i == 0 else bf(rnd(M, N, seed=31 + i))
        B = bf(rnd(M, K, seed=41 + i))
        refs.append(A.float().t() @ B.float())
        pairs.append((A.to(DEV) if i else wide.to(DEV)[:, :N], B.to(DEV)))
    arr, Mm, offs, total = o.tn_group(pairs)
    ns = o.tn_group_slices(arr, Mm)
    assert ns >= 1
    for nslices in sorted({ns, 3}):
        slabs = torch.full((nslices * total,), float("nan"), device=DEV)
        o.gemm_tn_grouped(arr, Mm, slabs, total, nslices)
        for (N, K), off, ref in zip(shapes, offs, refs):
            out = torch.zeros(N, K, device=DEV)
            o.reduce_slabs(out, slabs[off:], N * K, nslices, total, False)
            assert relerr(out, ref) < F32_TOL, (N, K, nslices)
    # the same launch with the column sums of dY (bias gradients) asked for on three of the five problems -- a multi-tile N,
    # a K of two tile columns (only the first computes them) and the partial tile -- and every product unchanged
    want = [True, False, False, True, True]
    arr, Mm, offs, total, cs = o.tn_group(pairs, colsum=want)
    assert [c is not None for c in cs] == want
    for nslices in sorted({ns, 3}):
        slabs = torch.full((nslices * total,), float("nan"), device=DEV)
        o.gemm_tn_grouped(arr, Mm, slabs, total, nslices)
        items = []
        outs = []
        for (N, K), off, cso, (A, _) in zip(shapes, offs, cs, pairs):
            out = torch.zeros(N, K, device=DEV)
            items.append((out, off, (1, N, N, 1, K, K)))
            outs.append(out)
            if cso is not None:
                ncrop = N - 8                         # a cropped destination (zero-padded models), accumulated into
                vec = torch.full((ncrop,), 2.0, device=DEV)
                items.append((vec, cso, (1, 1, 1, 1, ncrop, N)))
                outs.append((vec, A.float().sum(0)[:ncrop] + 2.0))
        o.reduce_slabs_group(items, slabs, nslices, total, True)
        k = 0
        for (N, K), ref, cso in zip(shapes, refs, cs):
            assert relerr(outs[k], ref) < F32_TOL, (N, K, nslices)
            k += 1
            if cso is not None:
                vec, vref = outs[k]
                assert relerr(vec, vref) < F32_TOL, ("colsum", N, nslices)
                k += 1


def test_gemm_tn_grouped_asymmetric():
    o = ops()
    M, N, K = 4096, 256, 384
    A = torch.zeros(M, N)
    B = torch.zeros(M, K)
    for m in range(M):
        A[m, (3 * m) % N] = 1.0
        B[m, (5 * m + 1) % K] = float((m % 61) + 1)
    ref = A.t() @ B
    Ad, Bd = bf(A).to(DEV), bf(B).to(DEV)          # (the problem array holds raw pointers: keep the operands alive)
    arr, Mm, offs, total = o.tn_group([(Ad, Bd)])
    slabs = torch.zeros(4 * total, device=DEV)
    o.gemm_tn_grouped(arr, Mm, slabs, total, 4)
    out = torch.zeros(N, K, device=DEV)
    o.reduce_slabs(out, slabs, N * K, 4, total, False)
    assert torch.equal(out.cpu(), ref)


def test_gemm_tn_asymmetric():
    o = ops()
    M, N, K = 32, 128, 128
    A = torch.zeros(M, N)
    B = torch.zeros(M, K)
    for m in range(M):
        A[m, (3 * m) % N] = 1.0
        B[m, (5 * m + 1) % K] = float(m + 1)
    ref = A.t() @ B
    for mode in (0, 1):
        out = torch.zeros(N, K, device=DEV)
        o.gemm_tn(bf(A).to(DEV), bf(B).to(DEV), out, mode=mode)
        assert torch.equal(out.cpu(), ref), f"mode {mode}"


# ---------------------------------------------------------------------------------------------
def test_embed_fwd_bwd():
    o = ops()
    V, D, n = 729, 128, 700
    E = rnd(V, D, seed=10)
    tok = torch.randint(0, V, (n,), generator=torch.Generator().manual_seed(11))
    tok[:40] = 5                                   # heavy collisions
    out = o.embed_fwd(tok.to(DEV), E.to(DEV))
    assert relerr(out, E[tok] * math.sqrt(D)) < BF16_TOL
    dX = bf(rnd(n, D, seed=12))
    ref = torch.zeros(V, D).index_add_(0, tok, dX.float()) * math.sqrt(D)
    dE = torch.ones(V, D, device=DEV)
    o.embed_bwd(tok.to(DEV), dX.to(DEV), dE, accumulate=True)
    assert relerr(dE - 1, ref) < 1e-5
    o.embed_bwd(tok.to(DEV), dX.to(DEV), dE, accumulate=False)
    assert relerr(dE, ref) < 1e-5


@pytest.mark.parametrize("D,ld", [(512, 512), (500, 512), (128, 136)])
def test_embed_bwd_token_order(D, ld):
    """The sorted-order kernel (stable argsort of the ids + segment offsets -> slabs -> reduce): against index_add, with
    dropout, collisions, ids outside the vocabulary (ignored) and a row stride wider than D; bit-identical run to run."""
    o = ops()
    V, n = 729, 9000
    tok = torch.randint(0, V, (n,), generator=torch.Generator().manual_seed(13))
    tok[:300] = 7
    tok[2000:5500] = 0            # a dominating id (the pad / start token of a real batch): runs across many chunks
    tok[5500:5600] = V - 1
    tok[300:310] = V + 5          # outside the vocabulary: contribute nothing
    tok[310:320] = -1
    dX = torch.zeros(n, ld, dtype=torch.bfloat16)
    dX[:, :D] = bf(rnd(n, D, seed=14))
    seed, p = 991, 0.1
    keep = o.dropout_keep_mask(seed, n * D, p).view(n, D).float()
    ok = ((tok >= 0) & (tok < V))
    ref = torch.zeros(V, D).index_add_(0, tok[ok], (dX[:, :D].float() * keep / (1 - p))[ok]) * math.sqrt(D)
    dE = torch.ones(V, D, device=DEV)
    order = o.token_order(tok.to(DEV), V)
    o.embed_bwd(tok.to(DEV), dX.to(DEV)[:, :D], dE, accumulate=True, drop_p=p, drop_seed=seed, order=order)
    assert relerr(dE - 1, ref) < 1e-5
    o.embed_bwd(tok.to(DEV), dX.to(DEV)[:, :D], dE, accumulate=False, order=order)
    assert relerr(dE, torch.zeros(V, D).index_add_(0, tok[ok], dX[:, :D].float()[ok]) * math.sqrt(D)) < 1e-5
    again = torch.empty_like(dE)
    o.embed_bwd(tok.to(DEV), dX.to(DEV)[:, :D], again, accumulate=False, order=order)
    assert torch.equal(dE, again)
    scan = torch.empty_like(dE)          # the scanning kernel (no order) agrees
    o.embed_bwd(tok.to(DEV), dX.to(DEV)[:, :D], scan, accumulate=False)
    assert relerr(scan, dE.cpu()) < 1e-5


@pytest.mark.parametrize("n", [1, 130, 9000, 65536])
def test_token_order_counting_sort_is_the_stable_argsort(n):
    """commu_token_order (histogram -> scan -> one wave per id) == torch.sort(stable=True) + searchsorted on the valid ids,
    with a dominating id, empty ids and ids outside the vocabulary (left out; the tail of perm reads row 0)."""
    o = ops()
    V = 729
    tok = torch.randint(0, V, (n,), generator=torch.Generator().manual_seed(n))
    if n >= 130:
        tok[n // 4: n // 2] = 0
        tok[5:9] = V + 3
        tok[9:12] = -7
        tok[tok == 11] = 12          # an id that never occurs
    perm, offs = o.token_order(tok.to(DEV), V)
    perm, offs = perm.cpu(), offs.cpu()
    valid = ((tok >= 0) & (tok < V)).nonzero().flatten()
    vals, order = torch.sort(tok[valid], stable=True)
    ref_perm = valid[order]
    ref_offs = torch.searchsorted(vals, torch.arange(V + 1))
    assert torch.equal(offs, ref_offs)
    assert torch.equal(perm[: valid.numel()], ref_perm)
    assert bool((perm[valid.numel():] == 0).all())


def test_out_of_range_ids_poison_the_loss():
    """A token / target id outside [0, V) (the reference raises IndexError) must neither read out of bounds nor
    pass silently: the embedding row and the row's nll come out NaN, the other rows are untouched."""
    o = ops()
    V, D, n = 729, 128, 64
    E = rnd(V, D, seed=10)
    tok = torch.randint(0, V, (n,), generator=torch.Generator().manual_seed(11))
    bad = tok.clone()
    bad[3], bad[17] = V, -1
    out = o.embed_fwd(bad.to(DEV), E.to(DEV)).float().cpu()
    ref = o.embed_fwd(tok.to(DEV), E.to(DEV)).float().cpu()
    good = torch.ones(n, dtype=torch.bool)
    good[3] = good[17] = False
    assert torch.isnan(out[~good]).all() and torch.equal(out[good], ref[good])
    logits = rnd(n, 736, seed=12).to(DEV)                  # fp32 logits, padded row pitch as in the model
    nll, _ = o.ce_fwd(logits, bad.to(DEV), V)
    nll_ref, _ = o.ce_fwd(logits, tok.to(DEV), V)
    assert torch.isnan(nll.cpu()[~good]).all() and torch.equal(nll.cpu()[good], nll_ref.cpu()[good])


@pytest.mark.parametrize("T,M,mem_len", [(4, 10, 10), (4, 10, 6), (7, 3, 9), (5, 0, 3)])
def test_mems_update_kernel(T, M, mem_len):
    """K9 against torch.cat(...)[beg:end] (model.py:507-538), including a memory tensor that is a strided view."""
    o = ops()
    Lp, B, Dp = 3, 5, 64
    hids = bf(rnd(Lp, T * B, Dp, seed=1)).to(DEV)
    big = bf(rnd(Lp, M + 2, B, Dp, seed=2)).to(DEV)
    mems = big[:, 2:]                                       # layer stride != M*B*Dp
    end = M + T
    beg = max(0, end - mem_len)
    ref = torch.cat([mems, hids.view(Lp, T, B, Dp)], 1)[:, beg:end]
    if beg >= M:
        pytest.skip("no copy in this case: the model returns a view of the hidden-state buffer")
    out = torch.full((Lp, end - beg, B, Dp), 7.0, device=DEV, dtype=torch.bfloat16)
    o.mems_update(hids, mems, out, beg)
    assert torch.equal(out, ref)


def test_reduce_slabs_crop():
    o = ops()
    rg, rt, rp, cg, ct, cp, ns = 6, 50, 64, 2, 100, 128, 3
    stride = rg * rp * cg * cp + 40
    slabs = rnd(ns, stride, seed=3).to(DEV)
    full = slabs[:, :rg * rp * cg * cp].sum(0).view(rg, rp, cg, cp)[:, :rt, :, :ct].reshape(rg * rt, cg * ct)
    dst = rnd(rg * rt, cg * ct, seed=4).to(DEV)
    want = dst + full
    o.reduce_slabs_crop(dst, slabs, (rg, rt, rp, cg, ct, cp), ns, stride, True)
    assert float((dst - want).abs().max()) < 1e-5
    o.reduce_slabs_crop(dst, slabs, (rg, rt, rp, cg, ct, cp), ns, stride, False)
    assert float((dst - full).abs().max()) < 1e-5


def test_reduce_slabs_group():
    """Every slab reduction of a grouped weight-gradient launch in one launch: plain, row-limited and cropped items."""
    o = ops()
    ns, total = 3, 64 * 96 + 768 * 32 + 6 * 64 * 2 * 128
    stride = total + 24
    slabs = rnd(ns, stride, seed=60).to(DEV)
    summed = slabs.sum(0)
    items, want = [], []
    d0 = rnd(64, 96, seed=61).to(DEV)                                   # plain [64, 96]
    items.append((d0, 0, (1, 64, 64, 1, 96, 96)))
    want.append(d0 + summed[:64 * 96].view(64, 96))
    off = 64 * 96
    d1 = rnd(729, 32, seed=62).to(DEV)                                  # first 729 of 768 rows
    items.append((d1, off, (1, 729, 768, 1, 32, 32)))
    want.append(d1 + summed[off:off + 768 * 32].view(768, 32)[:729])
    off += 768 * 32
    d2 = rnd(6 * 50, 2 * 100, seed=63).to(DEV)                          # cropped [6 x 50 of 64, 2 x 100 of 128]
    items.append((d2, off, (6, 50, 64, 2, 100, 128)))
    want.append(d2 + summed[off:off + 6 * 64 * 2 * 128].view(6, 64, 2, 128)[:, :50, :, :100].reshape(300, 200))
    want = [w.clone() for w in want]
    o.reduce_slabs_group(items, slabs, ns, stride, True)
    for (dst, _, _), w in zip(items, want):
        assert float((dst - w).abs().max()) < 1e-5


def test_posemb_distance_order():
    o = ops()
    K, D = 37, 64
    inv_freq = 1.0 / (10000 ** (torch.arange(0.0, D, 2.0) / D))
    out = o.posemb(inv_freq.to(DEV), K, D)
    ref = X.sinusoid_table(K, D).flip(0)          # reference row k is distance K-1-k
    assert float((out.float().cpu() - ref).abs().max()) < 8e-3     # bf16 rounding of values in [-1,1]


def test_posemb_clamp_len():
    """cfg.MODEL.clamp_len (model.py:581-582): positions above it share its row of the table."""
    o = ops()
    K, D, C = 41, 64, 9
    inv_freq = 1.0 / (10000 ** (torch.arange(0.0, D, 2.0) / D))
    out = o.posemb(inv_freq.to(DEV), K, D, clamp_len=C).float().cpu()
    ref = X.sinusoid_table(K, D, clamp_len=C).flip(0)
    assert float((out - ref).abs().max()) < 8e-3
    assert torch.equal(out[C:], out[C:C + 1].expand(K - C, D)) and not torch.equal(out[C - 1], out[C])
    out32 = o.posemb_f32(inv_freq.to(DEV), K, D, clamp_len=C).cpu()
    assert float((out32 - ref).abs().max()) < 4e-7


@pytest.mark.parametrize("rows,D", [(5, 64), (300, 128), (1000, 512), (130, 1024)])
def test_layernorm_fwd_bwd(rows, D):
    o = ops()
    z = bf(rnd(rows, D, seed=13) * 2 + 0.3)
    gamma, beta = 1 + 0.1 * rnd(D, seed=14), 0.1 * rnd(D, seed=15)
    y, mean, rstd = o.layernorm_fwd(z.to(DEV), gamma.to(DEV), beta.to(DEV))
    zf = z.float().requires_grad_(True)
    gp, bp = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    ref = X.layer_norm(zf, gp, bp)
    assert relerr(y, ref) < BF16_TOL
    assert relerr(mean, zf.detach().mean(-1)) < 1e-4
    dy = bf(rnd(rows, D, seed=16))
    ref.backward(dy.float())
    dz, part = o.layernorm_bwd(dy.to(DEV), z.to(DEV), mean, rstd, gamma.to(DEV))
    assert relerr(dz, zf.grad) < BF16_TOL
    sums = part.sum(0)
    assert relerr(sums[0], gp.grad) < 2e-3
    assert relerr(sums[1], bp.grad) < 2e-3
    assert relerr(sums[2], dz.float().sum(0)) < 2e-3


def test_colsum():
    o = ops()
    Xb = bf(rnd(777, 1000, seed=17))
    out = torch.ones(1000, device=DEV)
    o.colsum(Xb.to(DEV), out)
    assert relerr(out - 1, Xb.float().sum(0)) < 1e-3
    Xf = rnd(100, 40, seed=18)
    out = torch.zeros(40, device=DEV)
    o.colsum(Xf.to(DEV), out)
    assert relerr(out, Xf.sum(0)) < 1e-5
    # the shapes of the training step: a [rows, 729] view of 768-wide rows (pad columns hold garbage), large inputs
    # (two passes over a workspace), fp32 per-tile partials, odd column counts; and the sum is DETERMINISTIC (no atomics)
    big = bf(rnd(9000, 768, seed=21))
    big[:, 729:] = 1e4
    for X, cols in ((big.to(DEV)[:, :729], 729), (bf(rnd(70000, 1024, seed=22)).to(DEV), 1024),
                    (rnd(1024, 512, seed=23).to(DEV), 512), (rnd(5000, 500, seed=24).to(DEV), 500),
                    (rnd(3001, 203 * 4, seed=25).to(DEV)[:, :810], 810)):
        a, b = torch.zeros(cols, device=DEV), torch.zeros(cols, device=DEV)
        o.colsum(X, a)
        o.colsum(X, b)
        assert torch.equal(a, b)
        ref = X.double().sum(0).float()
        assert relerr(a, ref) < (2e-3 if X.dtype == torch.bfloat16 else 1e-5), (cols, relerr(a, ref))


def test_ce_fwd_bwd():
    o = ops()
    rows, V, ld = 300, 729, 768
    logits = torch.full((rows, ld), 1e4)           # pad columns hold garbage that must be ignored
    logits[:, :V] = rnd(rows, V, seed=19) * 3
    target = torch.randint(0, V, (rows,), generator=torch.Generator().manual_seed(20))
    nll, lse = o.ce_fwd(logits.to(DEV), target.to(DEV), V)
    lg = logits[:, :V].clone().requires_grad_(True)
    ref = torch.logsumexp(lg, -1) - lg.gather(1, target[:, None]).squeeze(1)
    assert relerr(nll, ref) < 1e-5
    g = rnd(rows, seed=21).abs()
    ref.backward(g)
    dl = o.ce_bwd(logits.to(DEV), target.to(DEV), lse, g.to(DEV), V)
    assert relerr(dl[:, :V], lg.grad) < BF16_TOL
    assert float(dl[:, V:].abs().max()) == 0.0


def test_masked_mean_and_grad():
    o = ops()
    n = 1000
    nll = rnd(n, seed=22).abs()
    target = torch.randint(0, 5, (n,), generator=torch.Generator().manual_seed(23))
    ws_sum, ws_cnt = torch.zeros(1, device=DEV), torch.zeros(1, device=DEV, dtype=torch.int32)
    out = torch.zeros(1, device=DEV)
    o.masked_mean(nll.to(DEV), target.to(DEV), 0, 0.25, ws_sum, ws_cnt, out)
    ref = nll[target != 0].mean() * 0.25
    assert abs(float(out) - float(ref)) < 1e-5
    g = torch.empty(n, device=DEV)
    o.loss_grad(target.to(DEV), 0, ws_cnt, 0.25, g)
    refg = (target != 0).float() * 0.25 / float((target != 0).sum())
    assert relerr(g, refg) < 1e-6


def test_adam_and_grad_norm_match_oracle():
    o = ops()
    n = 10007
    p0, g0 = rnd(n, seed=24), rnd(n, seed=25) * 0.1
    p = {"w": p0.clone()}
    st = X.adam_init(p)
    pd, gd = p0.to(DEV).clone(), g0.to(DEV)
    m, v = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    pb = torch.empty(n, device=DEV, dtype=torch.bfloat16)
    part, gn = torch.empty(256, device=DEV), torch.empty(1, device=DEV)
    for step in range(1, 4):
        total, coef = X.clip_coef([g0], 0.25)
        X.adam_update(p, {"w": g0 * coef}, st, 1e-2)
        o.grad_norm(gd, part, gn)
        assert abs(float(gn) - float(total)) < 1e-4 * float(total)
        o.adam_step(pd, gd, m, v, pb, 1e-2, step, gnorm=gn, clip=0.25)
        assert relerr(pd, p["w"]) < 1e-6
        assert relerr(pb, p["w"]) < 5e-3
    # the vector body moves 16 bytes per lane; a slice at an odd offset takes the element-wise body: same results
    p2 = {"w": p0[1:].clone()}
    st2 = X.adam_init(p2)
    X.adam_update(p2, {"w": g0[1:]}, st2, 1e-2)
    pd2, m2, v2 = p0.to(DEV).clone(), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    pb2 = torch.zeros(n + 1, device=DEV, dtype=torch.bfloat16)
    o.adam_step(pd2[1:], gd[1:], m2[1:], v2[1:], pb2[2:], 1e-2, 1)
    assert relerr(pd2[1:], p2["w"]) < 1e-6 and float(pd2[0]) == float(p0[0]) and float(m2[0]) == 0.0
    assert relerr(pb2[2:], p2["w"]) < 5e-3 and float(pb2[:2].abs().max()) == 0.0


def test_transposes():
    o = ops()
    x = rnd(100, 72, seed=26)
    assert torch.equal(o.transpose_to_bf16(x.to(DEV)).cpu(), bf(x).t())
    assert torch.equal(o.transpose_to_bf16(bf(x).to(DEV)).cpu(), bf(x).t())
    J, B, H, DH, W = 21, 3, 2, 32, 64
    src = bf(rnd(J * B, 3 * H * DH, seed=27))
    bias = rnd(H * DH, seed=28)
    out = o.transpose_heads(src.to(DEV)[:, H * DH:2 * H * DH], J, B, H, DH, W, off=5, bias=bias.to(DEV))
    ref = torch.zeros(B, H, DH, W)
    blk = (src[:, H * DH:2 * H * DH].float() + bias).view(J, B, H, DH)
    ref[:, :, :, 5:5 + J] = blk.permute(1, 2, 3, 0)
    assert relerr(out, ref) < BF16_TOL and float(out[..., :5].abs().max()) == 0


# ---------------------------------------------------------------------------------------------
def attn_reference(q, k, v, rd, u, vb, reset, T, M, B, H, DH, same_length, mem_len):
    """Oracle math (oracle/xl_ref.py) on fp32 copies of the bf16 inputs; rd is distance-indexed."""
    K = T + M
    r = rd.view(K, H, DH).flip(0)                  # reference row m is distance K-1-m
    S = X.rel_attention_scores(q.view(T, B, H, DH), k.view(K, B, H, DH), r, u.view(H, DH), vb.view(H, DH))
    S = S * (1.0 / math.sqrt(DH))
    mask = X.attn_mask(T, M, B, reset, same_length, mem_len)
    S = S.masked_fill(mask[:, None], float("-inf"))
    A = torch.softmax(S, dim=3)
    o = torch.einsum("bnij,jbnd->ibnd", A, v.view(K, B, H, DH)).reshape(T * B, H * DH)
    lse = torch.logsumexp(S, dim=3)
    return o, lse


ATTN_CASES = [
    # T, M, B, H, DH, same_length, mem_len, reset_col
    (64, 0, 2, 2, 64, False, 0, None),
    (128, 0, 1, 2, 32, False, 0, None),
    (12, 16, 3, 2, 32, False, 16, 1),
    (100, 60, 2, 2, 64, False, 64, 0),
    (16, 16, 2, 2, 64, True, 16, None),
    (70, 130, 2, 1, 64, True, 150, 1),
    (1, 77, 2, 2, 32, True, 4146, None),
    (200, 0, 1, 1, 64, False, 0, None),
    # causal-band mode of the backward (no same_length / reset): several GEMM row tiles, memory, B = 1 wedge
    (100, 60, 2, 2, 64, False, 64, None),
    (300, 40, 3, 1, 64, False, 40, None),
    (520, 0, 1, 2, 32, False, 0, None),
    (257, 31, 5, 1, 64, False, 31, None),
    # >= 8192 token rows: the backward takes the fused band pass (commu_relattn_bwd_band) -- causal band with and
    # without memory (poisoned scratch beyond the wedge), and the fully zero-initialised variant (masks)
    (128, 0, 64, 2, 64, False, 0, None),
    (160, 130, 52, 1, 64, False, 130, None),
    (96, 32, 96, 2, 64, True, 100, 3),
    # more than 1024 key positions: the fused band pass runs one pass per 512 distances (the reference's released
    # default shape tgt_len 128 / mem_len 1024, and a masked variant)
    (128, 1024, 64, 1, 64, False, 1024, None),
    (128, 1100, 64, 1, 64, True, 1100, 2),
    # more than 8 heads: the band pass takes 256 / H token-slice pairs per head (25 pairs = 50 slices of 3 steps for 128
    # steps: the last slices are empty; 16 pairs with memory)
    (128, 0, 64, 10, 64, False, 0, None),
    (64, 64, 128, 16, 64, False, 64, None),
    # memory as long as the segment with a reset column at a small batch (the shape of tests/test_configs_gpu.py's graph-step
    # case: the 32x32 key-stationary kernel as the default aborted there in round 6)
    (64, 64, 8, 4, 64, False, 64, 1),
]


def make_attn_inputs(T, M, B, H, DH, seed):
    K = T + M
    qkv = bf(rnd(K * B, 3 * H * DH, seed=seed) * 0.7)
    rd = bf(rnd(K, H * DH, seed=seed + 1) * 0.7)
    u, vb = rnd(H * DH, seed=seed + 2) * 0.3, rnd(H * DH, seed=seed + 3) * 0.3
    return qkv, rd, u, vb


@pytest.mark.parametrize("case", ATTN_CASES)
def test_relattn_fwd(case):
    o = ops()
    T, M, B, H, DH, sl, mem_len, rc = case
    K = T + M
    qkv, rd, u, vb = make_attn_inputs(T, M, B, H, DH, 30)
    reset = None
    if rc is not None:
        reset = torch.zeros(B, dtype=torch.bool)
        reset[rc] = True
    HD = H * DH
    qf, kf, vf = qkv[M * B:, :HD].float(), qkv[:, HD:2 * HD].float(), qkv[:, 2 * HD:].float()
    ref_o, ref_lse = attn_reference(qf, kf, vf, rd.float(), u, vb, reset, T, M, B, H, DH, sl, mem_len)
    g = qkv.to(DEV)
    rst = None if reset is None else reset.to(torch.uint8).to(DEV)
    out, lse, _ = o.relattn_fwd(g[M * B:, :HD], g[:, HD:2 * HD], g[:, 2 * HD:], rd.to(DEV), u.to(DEV), vb.to(DEV),
                                rst, T, M, B, H, DH, sl, mem_len)
    assert relerr(out, ref_o) < BF16_TOL
    assert float((lse.cpu() - ref_lse).abs().max()) < 6e-3   # bf16 (q+u), (q+v) operands


def test_relattn_bwd_delta_inside_the_query_kernel():
    """commu_attn_bwd_desc.o: the query-stationary kernel computes delta = rowsum(o . dO) itself and writes it for the
    key-stationary kernel (ops.DELTA_KERNEL = False) -- same gradients as with the separate commu_attn_delta launch."""
    o = ops()
    T, M, B, H, DH = 96, 40, 3, 2, 64
    K, HD = T + M, H * DH
    qkv, rd, u, vb = make_attn_inputs(T, M, B, H, DH, 50)
    g = qkv.to(DEV)
    q, k, v = g[M * B:, :HD], g[:, HD:2 * HD], g[:, 2 * HD:]
    dout = bf(rnd(T * B, HD, seed=51)).to(DEV)
    out, lse, qs = o.relattn_fwd(q, k, v, rd.to(DEV), u.to(DEV), vb.to(DEV), None, T, M, B, H, DH, False, M, save_q=True)
    res = []
    keep = o.DELTA_KERNEL
    try:
        for flag in (True, False):
            o.DELTA_KERNEL = flag
            dqkv = torch.zeros_like(g)
            drd = torch.zeros(K, HD, device=DEV)
            du, dvb = torch.zeros(HD, device=DEV), torch.zeros(HD, device=DEV)
            delta = o.relattn_bwd(q, k, v, rd.to(DEV), u.to(DEV), vb.to(DEV), None, T, M, B, H, DH, False, M, out, dout, lse, qs,
                                  dqkv[M * B:, :HD], dqkv[:, HD:2 * HD], dqkv[:, 2 * HD:], drd, du, dvb)
            torch.cuda.synchronize()
            res.append((dqkv.float().cpu(), drd.cpu(), du.cpu(), dvb.cpu()))
    finally:
        o.DELTA_KERNEL = keep
    for a_, b_ in zip(res[0], res[1]):
        assert relerr(b_, a_) < 2e-3


@pytest.mark.parametrize("store_p", [False, True, 3, 4, 5, "fwd_p", "fwd_p_kv3"],
                         ids=["recompute", "stored_p", "stored_p_kv3", "stored_p_q3", "stored_p_default", "forward_p", "forward_p_kv3"])
@pytest.mark.parametrize("case", ATTN_CASES)
def test_relattn_bwd(case, store_p):
    """store_p: the query-stationary kernel writes the probabilities it recomputes into a scratch buffer (poisoned with
    NaN here) and the key-stationary kernel reads them back (d_head 64); otherwise both recompute P from (q+u).k, the
    band product and lse.  forward_p: the FORWARD pass saved its probabilities (commu_relattn_fwd_save, buffer poisoned
    before the call: tiles the forward never visits hold NaN patterns) and the query-stationary kernel reads them instead of
    recomputing scores -- the training path for d_head 64.  All against autograd of the oracle."""
    o = ops()
    T, M, B, H, DH, sl, mem_len, rc = case
    fwd_p = isinstance(store_p, str)
    if fwd_p:
        store_p = 3 if store_p.endswith("kv3") else True
    if store_p and DH != 64:
        pytest.skip("stored probabilities: d_head 64 kernels only")
    K = T + M
    HD = H * DH
    qkv, rd, u, vb = make_attn_inputs(T, M, B, H, DH, 40)
    reset = None
    if rc is not None:
        reset = torch.zeros(B, dtype=torch.bool)
        reset[rc] = True
    leaf = qkv.float().requires_grad_(True)
    rdl, ul, vl = rd.float().requires_grad_(True), u.clone().requires_grad_(True), vb.clone().requires_grad_(True)
    ref_o, _ = attn_reference(leaf[M * B:, :HD], leaf[:, HD:2 * HD], leaf[:, 2 * HD:], rdl, ul, vl, reset, T, M, B,
                              H, DH, sl, mem_len)
    dout = bf(rnd(T * B, HD, seed=41))
    ref_o.backward(dout.float())

    g = qkv.to(DEV)
    rst = None if reset is None else reset.to(torch.uint8).to(DEV)
    q, k, v = g[M * B:, :HD], g[:, HD:2 * HD], g[:, 2 * HD:]
    dqkv = torch.zeros_like(g)
    drd = torch.zeros(K, HD, device=DEV)
    du, dvb = torch.zeros(HD, device=DEV), torch.zeros(HD, device=DEV)
    o.POISON_SCRATCH = True           # NaN in every scratch element the kernels are not supposed to read
    keep_save, o.FWD_SAVES_P = o.FWD_SAVES_P, fwd_p
    try:
        out, lse, qs = o.relattn_fwd(q, k, v, rd.to(DEV), u.to(DEV), vb.to(DEV), rst, T, M, B, H, DH, sl, mem_len,
                                     save_q=True, save_p=True)
    finally:
        o.FWD_SAVES_P = keep_save
    assert (qs[2] is not None) == fwd_p
    keep_flag, o.STORE_ATTN_P = o.STORE_ATTN_P, bool(store_p)
    # (stored_p_kv3: the key-stationary kernel of relattn_kv3.hip -- 32 keys per wave on the 32x32 MFMA -- and the block order
    #  of the scratch that goes with it; stored_p: the 16x16-layout kernel)
    # (stored_p_q3: both backward kernels on the 32x32 MFMA -- relattn_q3.hip stores P in its own accumulator order,
    #  relattn_kv3.hip transposes the blocks through LDS)
    # (stored_p_default: generation 0 = the 16x16 query-stationary kernel's whole-line block order read by relattn_kv3.hip)
    prev_kv = o.attn_bwd_kv_generation(store_p if store_p in (3, 4) else (0 if store_p == 5 else 2))
    try:
        o.relattn_bwd(q, k, v, rd.to(DEV), u.to(DEV), vb.to(DEV), rst, T, M, B, H, DH, sl, mem_len, out,
                      dout.to(DEV), lse, qs, dqkv[M * B:, :HD], dqkv[:, HD:2 * HD], dqkv[:, 2 * HD:], drd, du, dvb)
    finally:
        o.POISON_SCRATCH = False
        o.STORE_ATTN_P = keep_flag
        o.attn_bwd_kv_generation(prev_kv)
    gref = leaf.grad
    tol = 2.5e-2          # bf16 P/dS operands + bf16 outputs
    assert relerr(dqkv[M * B:, :HD], gref[M * B:, :HD]) < tol, "dq"
    assert relerr(dqkv[:, HD:2 * HD], gref[:, HD:2 * HD]) < tol, "dk"
    assert relerr(dqkv[:, 2 * HD:], gref[:, 2 * HD:]) < tol, "dv"
    assert float(dqkv[:M * B, :HD].abs().max()) == 0 if M > 0 else True
    assert relerr(drd, rdl.grad) < tol, "drd"
    assert relerr(du, ul.grad) < tol, "du"
    assert relerr(dvb, vl.grad) < tol, "dvb"
