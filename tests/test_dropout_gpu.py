"""K16 dropout on the GPU: every site uses the counter-based mask keep(seed, element index), so the
masks can be re-created on the CPU and injected into the oracle -- training-mode parity is checked
exactly (same masks), not just statistically.  Reference sites: model.py:166,168 (FFN), :210-211,
:337,349 (attention), :454,:585-586 (embedding, positions), :601 (final)."""
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import xl_ref as X  # noqa: E402

DEV = "cuda"


def relerr(a, b):
    a = torch.as_tensor(a).detach().float().cpu()
    b = torch.as_tensor(b).detach().float().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def bf(x):
    return x.to(torch.bfloat16)


def rnd(*shape, seed=0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


def test_mask_statistics_and_determinism():
    from commu_amd import ops
    n = 1 << 20
    for p in (0.1, 0.5):
        m = ops.dropout_keep_mask(ops.site_seed(123, 7), n, p)
        assert abs(float(m.float().mean()) - (1 - p)) < 3e-3
        m2 = ops.dropout_keep_mask(ops.site_seed(123, 8), n, p)
        agree = float((m == m2).float().mean())
        assert abs(agree - ((1 - p) ** 2 + p ** 2)) < 3e-3          # independent sites
    assert ops.site_seed(5, 1) != ops.site_seed(5, 2) != ops.site_seed(6, 1)
    # keyed hash: seeds that differ by a small integer must not give shifted copies of one mask
    a = ops.dropout_keep_mask(1000, n, 0.5)
    for d in (1, 2, 64, 4096):
        b = ops.dropout_keep_mask(1000 + d, n, 0.5)
        assert abs(float((a[d:] == b[:n - d]).float().mean()) - 0.5) < 5e-3, d
        assert abs(float((a == b).float().mean()) - 0.5) < 5e-3, d


def test_gemm_epilogue_dropout_exact_mask():
    from commu_amd import ops
    M, N, K, p, seed = 200, 264, 64, 0.25, 99991
    A, B = bf(rnd(M, K, seed=1)), bf(rnd(N, K, seed=2))
    bias, resid = rnd(N, seed=3), bf(rnd(M, N, seed=4))
    keep = ops.dropout_keep_mask(seed, M * N, p).view(M, N)
    ref = torch.relu(A.float() @ B.float().t() + bias) * keep / (1 - p) + resid.float()
    out = ops.gemm_nt(A.to(DEV), B.to(DEV), bias=bias.to(DEV), relu=True, resid=resid.to(DEV), out_f32=True,
                      drop_p=p, drop_seed=seed)
    assert relerr(out, ref) < 2e-3
    act = bf(rnd(M, N, seed=5))
    out = ops.gemm_nt(A.to(DEV), B.to(DEV), relu_mask=act.to(DEV), mask_scale=2.0, out_f32=True)
    assert relerr(out, (A.float() @ B.float().t()) * (act.float() > 0) * 2.0) < 2e-3


def test_embed_posemb_layernorm_dropout_exact_mask():
    from commu_amd import ops
    V, D, n, p = 729, 128, 300, 0.1
    E = rnd(V, D, seed=10)
    tok = torch.randint(0, V, (n,), generator=torch.Generator().manual_seed(11))
    keep = ops.dropout_keep_mask(77, n * D, p).view(n, D)
    out = ops.embed_fwd(tok.to(DEV), E.to(DEV), drop_p=p, drop_seed=77)
    assert relerr(out, E[tok] * math.sqrt(D) * keep / (1 - p)) < 1e-2
    dX = bf(rnd(n, D, seed=12))
    ref = torch.zeros(V, D).index_add_(0, tok, dX.float() * keep / (1 - p)) * math.sqrt(D)
    dE = torch.zeros(V, D, device=DEV)
    ops.embed_bwd(tok.to(DEV), dX.to(DEV), dE, accumulate=False, drop_p=p, drop_seed=77)
    assert relerr(dE, ref) < 1e-5
    K = 40
    inv_freq = 1.0 / (10000 ** (torch.arange(0.0, D, 2.0) / D))
    pe = ops.posemb(inv_freq.to(DEV), K, D, drop_p=p, drop_seed=5)
    kp = ops.dropout_keep_mask(5, K * D, p).view(K, D)
    assert float((pe.float().cpu() - X.sinusoid_table(K, D).flip(0) * kp / (1 - p)).abs().max()) < 1e-2
    rows = 130
    z = bf(rnd(rows, D, seed=13) + 0.2)
    gamma, beta = 1 + 0.1 * rnd(D, seed=14), 0.1 * rnd(D, seed=15)
    yd = torch.empty(rows, D, device=DEV, dtype=torch.bfloat16)
    y, mean, rstd = ops.layernorm_fwd(z.to(DEV), gamma.to(DEV), beta.to(DEV), y_drop=yd, drop_p=p, drop_seed=9)
    kl = ops.dropout_keep_mask(9, rows * D, p).view(rows, D)
    ref = X.layer_norm(z.float(), gamma, beta)
    assert relerr(y, ref) < 1.2e-2 and relerr(yd, ref * kl / (1 - p)) < 1.2e-2
    dy = bf(rnd(rows, D, seed=16))
    dzm = torch.empty(rows, D, device=DEV, dtype=torch.bfloat16)
    dz, part = ops.layernorm_bwd(dy.to(DEV), z.to(DEV), mean, rstd, gamma.to(DEV), dz_masked=dzm, drop_p=p, drop_seed=9)
    assert relerr(dzm, dz.float().cpu() * kl / (1 - p)) < 1e-2
    assert relerr(part.sum(0)[2], dzm.float().sum(0)) < 2e-3


def attn_keep(seed, B, H, T, K, p):
    """(keep mask, exact keep probability) of the attention kernels."""
    from commu_amd import ops
    return ops.attn_dropout_keep_mask(seed, B, H, T, K, p)


@pytest.mark.parametrize("fwd_gen", [0, 2], ids=["fwd_default", "fwd_16x16"])
@pytest.mark.parametrize("store_p", [False, True, 3, "fwd_p"], ids=["recompute", "stored_p", "stored_p_kv3", "forward_p"])
@pytest.mark.parametrize("case", [(64, 0, 2, 2, 64), (40, 24, 2, 2, 32), (130, 0, 1, 1, 64), (200, 70, 3, 2, 64)])
def test_attention_dropout_fwd_bwd_exact_mask(case, store_p, fwd_gen):
    """Forward and backward regenerate ONE mask (ops.attn_dropout_keep_mask): the default forward of d_head 64 is the
    transposed 32x32 kernel (relattn3.hip), the backward kernels are the 16x16 family -- different register layouts, the
    same keep decisions; fwd_16x16 forces the older forward.  forward_p (the training path): the forward saves its
    probabilities with the keep decision in the sign bit and the query-stationary backward kernel takes both from there."""
    from commu_amd import ops
    T, M, B, H, DH = case
    fwd_p = store_p == "fwd_p"
    if fwd_p:
        store_p = True
    if store_p and DH != 64:
        pytest.skip("stored probabilities: d_head 64 kernels only")
    if fwd_gen == 2 and (DH != 64 or fwd_p):
        pytest.skip("one forward kernel for this d_head / the 16x16 forward does not save probabilities")
    K, HD, p, seed = T + M, H * DH, 0.2, 424243
    g = torch.Generator().manual_seed(3)
    qkv = bf(torch.randn(K * B, 3 * HD, generator=g) * 0.7)
    rd = bf(torch.randn(K, HD, generator=g) * 0.7)
    u, vb = torch.randn(HD, generator=g) * 0.3, torch.randn(HD, generator=g) * 0.3
    keep, pkeep = attn_keep(seed, B, H, T, K, p)
    leaf = qkv.float().requires_grad_(True)
    rdl, ul, vl = rd.float().requires_grad_(True), u.clone().requires_grad_(True), vb.clone().requires_grad_(True)
    r = rdl.view(K, H, DH).flip(0)
    S = X.rel_attention_scores(leaf[M * B:, :HD].reshape(T, B, H, DH), leaf[:, HD:2 * HD].reshape(K, B, H, DH), r,
                               ul.view(H, DH), vl.view(H, DH)) / math.sqrt(DH)
    S = S.masked_fill(X.attn_mask(T, M, B, None, False, M)[:, None], float("-inf"))
    A = torch.softmax(S, 3) * keep / pkeep
    ref = torch.einsum("bnij,jbnd->ibnd", A, leaf[:, 2 * HD:].reshape(K, B, H, DH)).reshape(T * B, HD)
    dout = bf(torch.randn(T * B, HD, generator=g))
    ref.backward(dout.float())
    gq = qkv.to(DEV)
    q, k, v = gq[M * B:, :HD], gq[:, HD:2 * HD], gq[:, 2 * HD:]
    prev_gen = ops.attn_fwd_generation(fwd_gen)
    keep_save, ops.FWD_SAVES_P = ops.FWD_SAVES_P, fwd_p
    try:
        out, lse, qs = ops.relattn_fwd(q, k, v, rd.to(DEV), u.to(DEV), vb.to(DEV), None, T, M, B, H, DH, False, M,
                                       save_q=True, save_p=True, drop_p=p, drop_seed=seed)
    finally:
        ops.attn_fwd_generation(prev_gen)
        ops.FWD_SAVES_P = keep_save
    assert (qs[2] is not None) == fwd_p
    assert relerr(out, ref) < 1.5e-2
    dqkv = torch.zeros_like(gq)
    drd = torch.zeros(K, HD, device=DEV)
    du, dvb = torch.zeros(HD, device=DEV), torch.zeros(HD, device=DEV)
    keep_flag, ops.STORE_ATTN_P = ops.STORE_ATTN_P, bool(store_p)
    prev_kv = ops.attn_bwd_kv_generation(3 if store_p == 3 else 2)
    try:
        ops.relattn_bwd(q, k, v, rd.to(DEV), u.to(DEV), vb.to(DEV), None, T, M, B, H, DH, False, M, out, dout.to(DEV), lse,
                        qs, dqkv[M * B:, :HD], dqkv[:, HD:2 * HD], dqkv[:, 2 * HD:], drd, du, dvb, drop_p=p, drop_seed=seed)
    finally:
        ops.STORE_ATTN_P = keep_flag
        ops.attn_bwd_kv_generation(prev_kv)
    gref = leaf.grad
    tol = 3e-2
    assert relerr(dqkv[M * B:, :HD], gref[M * B:, :HD]) < tol
    assert relerr(dqkv[:, HD:2 * HD], gref[:, HD:2 * HD]) < tol
    assert relerr(dqkv[:, 2 * HD:], gref[:, 2 * HD:]) < tol
    assert relerr(drd, rdl.grad) < tol and relerr(du, ul.grad) < tol and relerr(dvb, vl.grad) < tol


@pytest.mark.parametrize("gen", [2, 3])
@pytest.mark.parametrize("case", [(64, 0, 2, 2), (130, 0, 1, 1), (200, 70, 3, 2), (300, 33, 2, 1)])
def test_attention_dropout_forward_generations_exact_mask(case, gen):
    """Both forward kernels of d_head 64 (commu_attn_fwd_generation 2: 16x16 layout, 3: transposed 32x32 layout) with
    attention dropout against the oracle with the mirror's mask injected; the mask's keep rate and the independence of the
    four elements of a 2x2 cell (they share the first hash round)."""
    from commu_amd import ops
    T, M, B, H = case
    DH = 64
    K, HD, p, seed = T + M, H * DH, 0.2, 99173
    g = torch.Generator().manual_seed(5)
    qkv = bf(torch.randn(K * B, 3 * HD, generator=g) * 0.7)
    rd = bf(torch.randn(K, HD, generator=g) * 0.7)
    u, vb = torch.randn(HD, generator=g) * 0.3, torch.randn(HD, generator=g) * 0.3
    keep, pkeep = ops.attn_dropout_keep_mask(seed, B, H, T, K, p)
    assert abs(float(keep.float().mean()) - pkeep) < 0.02
    kf = keep.float()[:, :, :T - T % 2, :K - K % 2]
    cell = [kf[:, :, a_::2, b_::2] for a_ in range(2) for b_ in range(2)]
    for x_ in range(4):
        for y_ in range(x_ + 1, 4):          # P(both kept) = pkeep^2 for any two elements of a cell
            assert abs(float((cell[x_] * cell[y_]).mean()) - pkeep * pkeep) < 0.03, (x_, y_)
    qf = qkv.float()
    r = rd.float().view(K, H, DH).flip(0)
    S = X.rel_attention_scores(qf[M * B:, :HD].reshape(T, B, H, DH), qf[:, HD:2 * HD].reshape(K, B, H, DH), r,
                               u.view(H, DH), vb.view(H, DH)) / math.sqrt(DH)
    S = S.masked_fill(X.attn_mask(T, M, B, None, False, M)[:, None], float("-inf"))
    A = torch.softmax(S, 3) * keep / pkeep
    ref = torch.einsum("bnij,jbnd->ibnd", A, qf[:, 2 * HD:].reshape(K, B, H, DH)).reshape(T * B, HD)
    gq = qkv.to(DEV)
    prev = ops.attn_fwd_generation(gen)
    try:
        out, lse, _ = ops.relattn_fwd(gq[M * B:, :HD], gq[:, HD:2 * HD], gq[:, 2 * HD:], rd.to(DEV), u.to(DEV), vb.to(DEV),
                                      None, T, M, B, H, DH, False, M, drop_p=p, drop_seed=seed)
        torch.cuda.synchronize()
    finally:
        ops.attn_fwd_generation(prev)
    assert relerr(out, ref) < 1.5e-2
    assert float((lse.cpu() - torch.logsumexp(S, 3)).abs().max()) < 6e-3          # the normaliser is the un-dropped sum


def test_model_train_mode_matches_oracle_with_same_masks(golden_dir):
    """Whole model in train() mode (dropout 0.1 / attention dropout 0.1) vs the oracle with the very
    same masks injected at the reference's nn.Dropout sites."""
    from commu_amd import ops
    from test_model_gpu import build_from_fixture
    z = np.load(os.path.join(golden_dir, "g1_train_mem.npz"))
    model, cfg = build_from_fixture(z)
    p_drop, p_att = 0.1, 0.15
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = p_drop
    for layer in model.layers:
        layer.dec_attn.dropatt.p = p_att
    model.train()
    L, H, D, DI, T, B, mem_len, sl = [int(x) for x in z["meta"][:8]]
    s = X.XLShape(L, H, D, DI)
    params = {k[3:]: torch.from_numpy(z[k]).clone().requires_grad_(True) for k in z.files
              if k.startswith("p::") and not k.endswith("inv_freq")}
    data, target = torch.from_numpy(z["data0"]), torch.from_numpy(z["target0"])
    data1, target1 = torch.from_numpy(z["data2"]), torch.from_numpy(z["target2"])
    reset = torch.from_numpy(z["reset2"])

    def make_drop(seed, Tq, Kk):
        site_id = {"emb": lambda li: 0, "pos": lambda li: 1, "final": lambda li: 2, "att": lambda li: 16 + 4 * li,
                   "o": lambda li: 17 + 4 * li, "hid": lambda li: 18 + 4 * li, "out": lambda li: 19 + 4 * li}

        def drop(site, x):
            kind, li = site
            ss = ops.site_seed(seed, site_id[kind](li))
            if kind == "att":
                keep, pkeep = attn_keep(ss, x.shape[0], x.shape[1], x.shape[2], x.shape[3], p_att)
                return x * keep / pkeep
            if kind == "pos":                       # the kernel's table is indexed by distance = reversed rows
                keep = ops.dropout_keep_mask(ss, x.numel(), p_drop).view(x.shape).flip(0)
                return x * keep / (1 - p_drop)
            keep = ops.dropout_keep_mask(ss, x.numel(), p_drop).view(x.shape)
            return x * keep / (1 - p_drop)
        return drop

    torch.manual_seed(1234)
    seeds = [int(torch.randint(0, 2 ** 31 - 1, (1,)).item()) for _ in range(2)]
    torch.manual_seed(1234)                         # the model draws the same base seeds
    model.zero_grad()
    loss0, mems = model(data.to(DEV), target.to(DEV), torch.zeros(B, dtype=torch.bool, device=DEV), None)
    loss1, _ = model(data1.to(DEV), target1.to(DEV), reset.to(DEV), mems)
    loss1[target1.to(DEV) != 0].float().mean().backward()

    o0, omems = X.forward_loss(params, s, data, target, torch.zeros(B, dtype=torch.bool), None, mem_len, bool(sl),
                               make_drop(seeds[0], T, T))
    o1, _ = X.forward_loss(params, s, data1, target1, reset, omems.detach(), mem_len, bool(sl),
                           make_drop(seeds[1], T, T + omems.shape[1]))
    assert float((loss0.detach().cpu() - o0.detach()).abs().max()) < 5e-2
    assert float((loss1.detach().cpu() - o1.detach()).abs().max()) < 5e-2
    assert relerr(mems, omems) < 2e-2
    grads = torch.autograd.grad(o1[target1 != 0].mean(), list(params.values()))
    cos = {}
    for (k, _), g in zip(params.items(), grads):
        got = dict(model.named_parameters())[k].grad.detach().float().cpu().flatten()
        cos[k] = float(torch.dot(got, g.flatten()) / (got.norm() * g.norm() + 1e-30))
    assert min(cos.values()) > 0.99, cos
    # eval mode is dropout free and deterministic
    model.eval()
    a, _ = model(data.to(DEV), target.to(DEV), None, None)
    b, _ = model(data.to(DEV), target.to(DEV), None, None)
    assert torch.equal(a, b)
