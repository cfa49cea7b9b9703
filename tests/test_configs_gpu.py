"""Parity at the shapes BASELINE.json names (configs[0], [1], [4]) against the CPU oracle on seeded inputs:
per-token loss, new memory, and every gradient tensor.  Batches are kept small so the fp32 oracle finishes
in seconds; the full-batch properties (finite loss, loss decreases over optimiser steps) are checked at the
bench batch size without the oracle.

Tolerances are the bf16 ones of tests/test_model_gpu.py (operands bf16, accumulation fp32).
"""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import xl_ref as X  # noqa: E402

DEV = "cuda"

CONFIGS = [
    # tag, L, H, D, DI, T, mem_len, B, segments
    ("cfg0_L2_D128_T256", 2, 4, 128, 256, 256, 0, 3, 1),          # configs[0]: reference train.py plumbing case
    ("cfg1_L6_D512_T1024", 6, 8, 512, 1024, 1024, 0, 2, 1),       # configs[1]/[2]: the bench shape
    ("cfg4_L2of12_D1024_T2048_M2048", 2, 16, 1024, 2048, 2048, 2048, 1, 2),   # configs[4]: shape of one layer pair
    ("cfg4_L12_D1024_T64_M64_full_depth", 12, 16, 1024, 2048, 64, 64, 2, 2),  # configs[4]: all 12 layers, short segments
    # the reference's released default (config_helper.py:7-10,23-24): d_model 500, 10 heads of 50, d_inner 1000 --
    # zero-padded to 512 / 64 / 1024 inside the kernels
    ("default_L6_D500_dh50_T128_M1024", 6, 10, 500, 1000, 128, 1024, 2, 3),
]


def build(L, H, D, DI, T, mem_len, seed):
    from commu_amd.model.config_helper import get_cfg
    from commu_amd.model.dataset import BaseVocab
    from commu_amd.model.model import MemTransformerLM
    cfg = get_cfg(num_layers=L, num_heads=H, units=D, inner_size=DI, tgt_length=T, mem_length=mem_len,
                  dropout=0.0, attention_dropout=0.0)
    s = X.XLShape(L, H, D, DI)
    params = X.init_params(s, seed, std=0.02)
    for k in params:                      # give the zero-initialised biases some signal
        if k.endswith(".bias"):
            params[k] = 0.02 * torch.randn(params[k].shape, generator=torch.Generator().manual_seed(seed + 7))
    model = MemTransformerLM(cfg, BaseVocab())
    sd = {k: v.clone() for k, v in params.items()}
    sd["crit.out_layers.0.weight"] = sd["word_emb.emb_layers.0.weight"]
    missing = model.load_state_dict(sd, strict=False)
    assert not [m for m in missing.missing_keys if "inv_freq" not in m], missing
    return model.to(DEV), cfg, s, params


@pytest.mark.parametrize("case", CONFIGS, ids=[c[0] for c in CONFIGS])
def test_config_shape_loss_and_grads_vs_oracle(case):
    tag, L, H, D, DI, T, mem_len, B, nseg = case
    model, cfg, s, params = build(L, H, D, DI, T, mem_len, seed=11)
    model.eval()
    g = torch.Generator().manual_seed(5)
    oparams = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    mems, omems = None, None
    model.zero_grad()
    for seg in range(nseg):
        data = torch.randint(1, 729, (T, B), generator=g)
        target = torch.randint(1, 729, (T, B), generator=g)
        target[-5:, 0] = 0                                   # some pads
        reset = torch.zeros(B, dtype=torch.bool)
        loss, mems = model(data.to(DEV), target.to(DEV), reset.to(DEV), mems)
        nll, omems = X.forward_loss(oparams, s, data, target, reset, omems, mem_len, False)
        err = (loss.detach().float().cpu() - nll.detach()).abs()
        assert float(err.max()) < 6e-2 and float(err.mean()) < 8e-3, (tag, seg, float(err.max()), float(err.mean()))
        if mem_len > 0:
            assert tuple(mems.shape) == tuple(omems.shape)
            d = (mems.float().cpu() - omems.detach()).abs().max() / omems.detach().abs().max()
            assert float(d) < 3e-2, (tag, seg, float(d))
            omems = omems.detach()
        X.masked_mean_loss(nll, target).backward()
        loss[target.to(DEV) != 0].float().mean().backward()
    cos = {}
    for name, p in model.named_parameters():
        if name not in oparams or oparams[name].grad is None:
            continue
        a, b = p.grad.detach().float().cpu().flatten(), oparams[name].grad.flatten()
        cos[name] = float(torch.dot(a, b) / (a.norm() * b.norm() + 1e-30))
    assert len(cos) >= 3 + 11 * L
    worst = sorted(cos.items(), key=lambda kv: kv[1])[:3]
    print(f"{tag}: lowest gradient cosines {[(k, round(v, 5)) for k, v in worst]}")
    # measured: >= 0.9995 everywhere except the first FFN Linear (bf16 pre-activations flip a few ReLU gates against the
    # fp32 oracle: 0.995-0.999; checked to 6e-2 with the build's gates injected in test_model_gpu) -- and, twelve layers
    # deep, what those flips do to every gradient below them (0.9957-0.9965)
    rest = 0.995 if L >= 12 else 0.997
    bad = {k: v for k, v in cos.items() if v < (0.993 if "pos_ff.CoreNet.0" in k else rest)}
    assert not bad, (tag, bad)


def test_gelu_ffn_option_vs_oracle():
    """model.ffn_activation = "gelu" (a non-default option: BASELINE.json's north star names a GELU-FFN, the reference's FFN
    is ReLU): element-wise kernels against torch.nn.functional.gelu with the dropout mask injected, then per-token loss and
    every gradient of a 2-layer model against the oracle with the same activation.  Never used for parity claims."""
    from commu_amd import ops
    g = torch.Generator().manual_seed(3)
    z = (torch.randn(300, 200, generator=g) * 1.5).to(torch.bfloat16)
    dy = torch.randn(300, 200, generator=g).to(torch.bfloat16)
    p, seed = 0.25, 4711
    keep = ops.dropout_keep_mask(seed, 300 * 200, p).view(300, 200).float()
    zl = z.float().requires_grad_(True)
    ref = torch.nn.functional.gelu(zl) * keep / (1 - p)
    ref.backward(dy.float())
    zd = torch.zeros(300, 208, dtype=torch.bfloat16, device=DEV)
    zd[:, :200] = z.to(DEV)
    out = ops.gelu_fwd(zd[:, :200], drop_p=p, drop_seed=seed)
    dyd = torch.zeros(300, 208, dtype=torch.bfloat16, device=DEV)
    dyd[:, :200] = dy.to(DEV)
    dz = ops.gelu_bwd(dyd[:, :200], zd[:, :200], drop_p=p, drop_seed=seed)
    assert float((out.float().cpu() - ref.detach()).abs().max()) < 2e-2 * float(ref.abs().max())
    assert float((dz.float().cpu() - zl.grad).abs().max()) < 2e-2 * float(zl.grad.abs().max())

    L, H, D, DI, T, B = 2, 4, 128, 256, 96, 3
    model, cfg, s, params = build(L, H, D, DI, T, 0, seed=5)
    model.ffn_activation = "gelu"
    model.eval()
    data = torch.randint(1, 729, (T, B), generator=g)
    target = torch.randint(1, 729, (T, B), generator=g)
    reset = torch.zeros(B, dtype=torch.bool)
    oparams = {k: v.clone().requires_grad_(True) for k, v in params.items()}

    class Gelu:
        activation = "gelu"

        def __call__(self, site, x):
            return x
    model.zero_grad()
    loss, _ = model(data.to(DEV), target.to(DEV), reset.to(DEV), None)
    nll, _ = X.forward_loss(oparams, s, data, target, reset, None, 0, False, drop=Gelu())
    err = (loss.detach().float().cpu() - nll.detach()).abs()
    assert float(err.max()) < 6e-2 and float(err.mean()) < 8e-3
    relu_nll, _ = X.forward_loss(params, s, data, target, reset, None, 0, False)
    assert float((relu_nll - nll.detach()).abs().max()) > 1e-3          # (the option really changes the function)
    nll.mean().backward()
    loss.float().mean().backward()
    for name, prm in model.named_parameters():
        if name in oparams and oparams[name].grad is not None:
            a, b = prm.grad.detach().float().cpu().flatten(), oparams[name].grad.flatten()
            assert float(torch.dot(a, b) / (a.norm() * b.norm() + 1e-30)) > 0.995, name


def test_bench_shape_full_batch_columns_vs_oracle():
    """The headline shape AT THE BENCH BATCH (64 columns x 1024 tokens, eval mode): the paths that depend on the grid
    size -- grouped / eight-phase GEMMs over 65 536 rows, the fused band pass, the XCD tile order, the 256-row attention
    workgroups over 512 (batch, head) pairs -- against the oracle on three of the 64 columns (columns are independent in
    the forward pass: the oracle runs them alone), and the same three columns against a 3-column run of the build."""
    L, H, D, DI, T, B = 6, 8, 512, 1024, 1024, 64
    model, cfg, s, params = build(L, H, D, DI, T, 0, seed=21)
    model.eval()
    g = torch.Generator().manual_seed(17)
    data = torch.randint(1, 729, (T, B), generator=g)
    target = torch.randint(1, 729, (T, B), generator=g)
    target[-9:, 5] = 0                                           # some pads in a checked column
    reset = torch.zeros(B, dtype=torch.bool)
    cols = [0, 5, 63]
    with torch.no_grad():
        loss, _ = model(data.to(DEV), target.to(DEV), reset.to(DEV), None)
        loss = loss.float().cpu()
        nll, _ = X.forward_loss(params, s, data[:, cols], target[:, cols], reset[cols], None, 0, False)
        small, _ = model(data[:, cols].contiguous().to(DEV), target[:, cols].contiguous().to(DEV),
                         reset[cols].to(DEV), None)
    err = (loss[:, cols] - nll).abs()
    assert float(err.max()) < 6e-2 and float(err.mean()) < 8e-3, (float(err.max()), float(err.mean()))
    assert bool(torch.isfinite(loss).all())
    d = (loss[:, cols] - small.float().cpu()).abs()               # same kernels, other grids: bf16 rounding only
    assert float(d.max()) < 3e-2 and float(d.mean()) < 3e-3, (float(d.max()), float(d.mean()))


def test_bench_shape_full_batch_gradients_vs_oracle():
    """The BACKWARD pass at the bench batch against the oracle.  With every target outside three columns set to pad (0),
    loss[target != 0].mean() depends on those three columns only and -- the forward being column-independent -- EVERY
    parameter gradient of the 64-column run must equal the oracle's gradient of the 3-column run (train.py:148-155), while
    all kernels run their bench-size grids: grouped weight-gradient GEMMs over 65 536 rows, the band pass over 512 (batch,
    head) pairs, the side-stream ordering, the embedding scatter over 65 536 tokens.  Bounds: those of
    test_config_shape_loss_and_grads_vs_oracle."""
    L, H, D, DI, T, B = 6, 8, 512, 1024, 1024, 64
    model, cfg, s, params = build(L, H, D, DI, T, 0, seed=23)
    model.eval()
    g = torch.Generator().manual_seed(19)
    data = torch.randint(1, 729, (T, B), generator=g)
    full = torch.randint(1, 729, (T, B), generator=g)
    cols = [0, 31, 63]
    target = torch.zeros_like(full)
    target[:, cols] = full[:, cols]
    target[-7:, 31] = 0                                          # some pads inside a live column too
    reset = torch.zeros(B, dtype=torch.bool)
    model.zero_grad()
    loss, _ = model(data.to(DEV), target.to(DEV), reset.to(DEV), None)
    loss[target.to(DEV) != 0].float().mean().backward()
    oparams = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    nll, _ = X.forward_loss(oparams, s, data[:, cols], target[:, cols], reset[cols], None, 0, False)
    X.masked_mean_loss(nll, target[:, cols]).backward()
    err = (loss.detach().float().cpu()[:, cols] - nll.detach()).abs()
    assert float(err.max()) < 6e-2 and float(err.mean()) < 8e-3, (float(err.max()), float(err.mean()))
    cos, rel = {}, {}
    for name, prm in model.named_parameters():
        if name not in oparams or oparams[name].grad is None:
            continue
        a, b = prm.grad.detach().float().cpu().flatten(), oparams[name].grad.flatten()
        assert bool(torch.isfinite(a).all()), name
        cos[name] = float(torch.dot(a, b) / (a.norm() * b.norm() + 1e-30))
        rel[name] = float((a - b).norm() / (b.norm() + 1e-30))
    assert len(cos) >= 3 + 11 * L
    worst = sorted(cos.items(), key=lambda kv: kv[1])[:3]
    print(f"bench batch: lowest gradient cosines {[(k, round(v, 5)) for k, v in worst]}; "
          f"largest relative errors {[(k, round(v, 4)) for k, v in sorted(rel.items(), key=lambda kv: -kv[1])[:3]]}")
    # measured (round 6): cosines >= 0.9994, relative errors <= 0.036 (worst: the first FFN Linear of layer 0, then r_net.weight)
    bad = {k: v for k, v in cos.items() if v < 0.998}
    assert not bad, bad
    # (no column but the three may have contributed: the norms agree, not only the directions)
    badn = {k: v for k, v in rel.items() if v > 0.06}
    assert not badn, badn


def test_bench_shape_full_batch_trains():
    """configs[1] at the bench batch (64 x 1024 tokens): finite loss that goes down over optimiser steps."""
    from commu_amd.model.config_helper import get_cfg
    from commu_amd.model.dataset import BaseVocab, synthetic_batch
    from commu_amd.train import Trainer, build_model
    cfg = get_cfg(num_layers=6, num_heads=8, units=512, inner_size=1024, tgt_length=1024, mem_length=0,
                  batch_size=64, batch_chunk=1, dropout=0.1, attention_dropout=0.1)
    model = build_model(cfg, BaseVocab(), torch.device(DEV), seed=3)
    model.train()
    tr = Trainer(model, cfg, num_gpus=1)
    d, t, r, n = synthetic_batch(1024, 64, torch.device(DEV), seed=9)
    losses = [float(tr.step(d, t, r, n)) for _ in range(6)]
    assert all(math.isfinite(x) for x in losses), losses
    assert abs(losses[0] - math.log(729)) < 0.3, losses           # random init: ~uniform over the vocabulary
    assert losses[-1] < losses[1], losses                         # lr(0) = 0 (quirk Q7): step 0 does not move


@pytest.mark.parametrize("fp8", [False, True], ids=["bf16", "fp8_forward"])
def test_cfg5_full_12_layers_with_memory_trains(fp8):
    """configs[4] in full (12 layers, d_model 1024, 16 heads, FFN 2048, tgt_len 2048, mem_len 2048; bf16 operands, and --
    as BASELINE.json names the config -- with the layers' forward Linear products in MX-fp8, `model.fp8_forward`):
    the oracle is too slow at this size, so the size-independent properties: the loss at random init is ~log V,
    every step is finite, the memory is carried ([13, 2048, B, 1024], equal to the layer outputs of the step that
    wrote it) and the loss goes down over optimiser steps on a repeated batch."""
    from commu_amd.model.config_helper import get_cfg
    from commu_amd.model.dataset import BaseVocab, synthetic_batch
    from commu_amd.train import Trainer, build_model
    B = 4
    cfg = get_cfg(num_layers=12, num_heads=16, units=1024, inner_size=2048, tgt_length=2048, mem_length=2048,
                  batch_size=B, batch_chunk=1, dropout=0.1, attention_dropout=0.1)
    model = build_model(cfg, BaseVocab(), torch.device(DEV), seed=3)
    model.train()
    model.fp8_forward = fp8
    tr = Trainer(model, cfg, num_gpus=1)
    d, t, r, n = synthetic_batch(2048, B, torch.device(DEV), seed=9)
    r = torch.zeros_like(r)                                       # keep the memory: attention spans 4096 positions
    losses = []
    for i in range(6):
        losses.append(float(tr.step(d, t, r, n)))
        mems = tr.mems[0]
        assert tuple(mems.shape) == (13, 2048, B, 1024) and mems.dtype == torch.bfloat16
        assert bool(torch.isfinite(mems.float()).all())
    assert all(math.isfinite(x) for x in losses), losses
    # (at d_model 1024 the tied embedding makes the INPUT token's own logit ~ |E[tok]|^2 sqrt(D)/std ~ 20 at init: the
    #  loss starts well above log V; the oracle agrees at this width, see the cfg4_* cases above)
    assert math.log(729) - 0.3 < losses[0] < 12.0, losses
    assert losses[-1] < losses[1] - 0.05, losses
    for p in model.parameters():
        assert p.grad is None or bool(torch.isfinite(p.grad).all())


def test_default_config_generate_and_kv_cache_decode_dh50():
    """forward_generate and the K/V-cache decode step at the released head shape (d_head 50 -> padded 64):
    logits vs the oracle, and the cached single-token step vs the full forward."""
    from commu_amd.generate import DecodeState
    L, H, D, DI = 2, 10, 500, 1000
    model, cfg, s, params = build(L, H, D, DI, 1, 4146, seed=21)
    model.eval()
    model.same_length = True
    model.reset_length(1, 4146)
    g = torch.Generator().manual_seed(8)
    B = 3
    ctx = torch.randint(1, 729, (12, B), generator=g)
    with torch.no_grad():
        ref_logits, ref_mems = X.forward_generate(params, s, ctx, None, 4146, True)
    logits, mems = model.forward_generate(ctx.to(DEV), None)
    assert tuple(mems.shape) == tuple(ref_mems.shape)
    err = (logits.float().cpu() - ref_logits).abs().max() / ref_logits.abs().max()
    assert float(err) < 2e-2, float(err)
    assert float((mems.float().cpu() - ref_mems).abs().max() / ref_mems.abs().max()) < 3e-2
    # a second call fed with the returned (view of padded) memory
    nxt = torch.randint(1, 729, (1, B), generator=g)
    with torch.no_grad():
        ref2, _ = X.forward_generate(params, s, nxt, ref_mems, 4146, True)
    l2, _ = model.forward_generate(nxt.to(DEV), mems)
    assert float((l2.float().cpu() - ref2).abs().max() / ref2.abs().max()) < 2e-2
    # K/V cache: prefill 12 tokens, one cached step with the 13th
    st = DecodeState(model, B, 64)
    st.prefill(ctx.to(DEV))
    ones = torch.ones(B, dtype=torch.uint8, device=DEV)
    lg = st.step(nxt[0].to(DEV), ones, ones)[:, :729]
    assert float((lg.float().cpu() - ref2[0]).abs().max() / ref2.abs().max()) < 2e-2
    assert torch.equal(lg.argmax(1).cpu(), ref2[0].argmax(1)) or float(
        (ref2[0].topk(2, dim=1).values[:, 0] - ref2[0].topk(2, dim=1).values[:, 1]).min()) < 0.05


def test_evaluate_same_length_memory_window_vs_oracle():
    """Trainer.evaluate (train.py:74-110): same_length masks with the longer EVALUATE memory; the memory
    overflows its window after the second segment, so the same_length shift (model.py:549-568) is exercised."""
    from commu_amd.train import Trainer
    L, H, D, DI, T, B = 2, 2, 100, 136, 48, 3
    model, cfg, s, params = build(L, H, D, DI, T, 0, seed=31)
    cfg.defrost() if hasattr(cfg, "defrost") else None
    cfg.EVALUATE.tgt_length, cfg.EVALUATE.mem_length = T, 80
    g = torch.Generator().manual_seed(12)
    segs = []
    for i in range(4):
        data = torch.randint(1, 729, (T, B), generator=g)
        target = torch.randint(1, 729, (T, B), generator=g)
        target[-3:, 1] = 0
        segs.append((data, target, i == 2, int((target != 0).sum())))       # third segment resets the memory

    def eval_iter():
        for d, t, r, n in segs:
            yield d.to(DEV), t.to(DEV), r, n
    tr = Trainer(model, cfg)
    tok, nll = tr.evaluate(eval_iter)
    ref_tok, ref_nll, omems = 0, 0.0, None
    with torch.no_grad():
        for d, t, r, n in segs:
            if r:
                omems = None
            loss, omems = X.forward_loss(params, s, d, t, None, omems, 80, True)
            ref_nll += n * float(loss[t != 0].mean())
            ref_tok += n
    assert tok == ref_tok
    assert abs(nll / tok - ref_nll / ref_tok) < 6e-3, (nll / tok, ref_nll / ref_tok)
    assert model.training and model.mem_len == cfg.TRAIN.mem_length and model.same_length == cfg.MODEL.same_length


def test_training_learns_a_deterministic_rule():
    """End-to-end sanity of forward + backward + clip + Adam + LR schedule: x[t+1] = x[t] + 3 (mod 700) is learned
    to near-zero loss in 150 optimiser steps (a wrong gradient anywhere stalls near ln 729 = 6.59)."""
    from commu_amd.model.config_helper import get_cfg
    from commu_amd.model.dataset import BaseVocab
    from commu_amd.train import Trainer, build_model
    dev = torch.device(DEV)
    T, B = 64, 16
    cfg = get_cfg(num_layers=2, num_heads=4, units=128, inner_size=256, tgt_length=T, mem_length=T, batch_size=B,
                  batch_chunk=1, dropout=0.0, attention_dropout=0.0)
    tr = Trainer(build_model(cfg, BaseVocab(), dev, seed=1), cfg)
    g = torch.Generator().manual_seed(0)
    first = last = None
    for s in range(150):
        start = torch.randint(2, 700, (1, B), generator=g)
        seq = (start + 3 * torch.arange(T + 1)[:, None]) % 700 + 2
        loss = tr.step(seq[:-1].to(dev), seq[1:].to(dev), torch.ones(B, dtype=torch.bool, device=dev), T * B)
        last = float(loss.detach())
        first = last if first is None else first
    assert first > 6.0 and last < 0.1, (first, last)


@pytest.mark.parametrize("mem_len,T,B", [(0, 256, 16), (256, 256, 16), (128, 256, 16)],
                         ids=["nomem", "mem_eq_tgt", "mem_lt_tgt"])
def test_grouped_weight_gradients_match_the_per_gemm_path(mem_len, T, B, monkeypatch):
    """The production weight-gradient path (grouped eight-phase TN launch + ONE grouped slab reduction, side streams)
    against the per-GEMM fallback (`COMMU_TN8_OFF`) at T*B >= 4096, where the grouped kernel is eligible -- including
    mem_len == tgt_len, where the memory-side k|v gradient has the token count of the qkv gradient and both add into
    the same rows of qkv_net.weight.grad (they must not share one reduce launch).  Same seeds, dropout off: every
    gradient tensor agrees to fp32-reduction-order noise, on repeated runs (the old race was nondeterministic)."""
    L, H, D, DI = 2, 4, 256, 512
    model, cfg, s, params = build(L, H, D, DI, T, mem_len, seed=17)
    model.eval()
    g = torch.Generator().manual_seed(23)
    segs = [(torch.randint(1, 729, (T, B), generator=g), torch.randint(1, 729, (T, B), generator=g)) for _ in range(2)]

    def grads():
        model.zero_grad()
        mems = None
        for data, target in segs:
            loss, mems = model(data.to(DEV), target.to(DEV), torch.zeros(B, dtype=torch.bool, device=DEV), mems)
            loss.float().mean().backward()
        torch.cuda.synchronize()
        return {n: p.grad.detach().float().clone() for n, p in model.named_parameters() if p.grad is not None}

    monkeypatch.setenv("COMMU_TN8_OFF", "1")
    ref = grads()
    monkeypatch.delenv("COMMU_TN8_OFF")
    from commu_amd import ops
    dY = torch.zeros(T * B, 3 * D, device=DEV, dtype=torch.bfloat16)
    Xa = torch.zeros(T * B, D, device=DEV, dtype=torch.bfloat16)
    arr, Mtok, _, _ = ops.tn_group([(dY, Xa)])
    assert ops.tn_group_slices(arr, Mtok) > 0, "the grouped kernel must be eligible at this shape"
    for rep in range(3):
        got = grads()
        for n in ref:
            scale = float(ref[n].abs().max()) + 1e-12
            err = float((got[n] - ref[n]).abs().max()) / scale
            assert err < 2e-3, (mem_len, rep, n, err)


@pytest.mark.parametrize("variant", ["nomem_chunk1", "mem_chunk2", "padded_dh50"])
def test_graph_step_matches_eager_step(variant):
    """Trainer(graph=True): the optimiser step replayed from two hipGraphs (forward + backward of every micro-batch;
    clip + Adam + weight shadows) against the eager step, same seeds: dropout ON with the same base seeds and the same
    per-step salt written to device memory (eager kernels read the salt too), LR schedule through the device scalars.
    Parameters after 6 steps agree to fp32 summation-order noise (bias / LayerNorm column sums end in atomics), the
    losses of every step likewise, and the XL memory is carried through the graph's static buffers."""
    from commu_amd import ops
    from commu_amd.model.config_helper import get_cfg
    from commu_amd.model.dataset import BaseVocab, synthetic_batch
    from commu_amd.train import Trainer, build_model
    dev = torch.device(DEV)
    if variant == "nomem_chunk1":
        kw = dict(num_layers=2, num_heads=4, units=256, inner_size=512, tgt_length=128, mem_length=0, batch_size=8, batch_chunk=1)
    elif variant == "mem_chunk2":
        kw = dict(num_layers=2, num_heads=4, units=256, inner_size=512, tgt_length=64, mem_length=64, batch_size=8, batch_chunk=2)
    else:
        kw = dict(num_layers=2, num_heads=2, units=100, inner_size=200, tgt_length=64, mem_length=64, batch_size=4, batch_chunk=1)
    cfg = get_cfg(dropout=0.1, attention_dropout=0.1, **kw)
    T, B = kw["tgt_length"], kw["batch_size"]
    batches = [synthetic_batch(T, B, dev, seed=50 + i, reset_prob=0.2 if kw["mem_length"] else 0.0) for i in range(6)]
    salts = [1234567 + 977 * i for i in range(6)]
    out = {}
    for mode in ("eager", "graph"):
        model = build_model(cfg, BaseVocab(), dev, seed=3)
        model.train()
        model.fixed_drop_seed = 4242          # same base seeds in both runs; the per-step salt makes the masks differ by step
        tr = Trainer(model, cfg, num_gpus=1, graph=(mode == "graph"))
        salt = torch.zeros(1, device=dev, dtype=torch.int32)
        losses, replayed = [], 0
        for i, b in enumerate(batches):
            if tr.graph_mode and tr._graph_ready(b[0], b[2]):
                tr.salt_source = lambda i=i: salts[i]          # written to device memory by _graph_step
                losses.append(float(tr.step(*b)))
                replayed += 1
            else:
                salt.fill_(salts[i])
                ops.set_seed_salt(salt)                        # eager kernels read the salt as well
                try:
                    losses.append(float(tr.step(*b)))
                finally:
                    ops.set_seed_salt(None)
        torch.cuda.synchronize()
        if mode == "graph":
            assert tr.graph_failed is None, tr.graph_failed
            assert tr._graphs is not None and replayed >= 2
        out[mode] = (losses, {n: p.detach().float().cpu().clone() for n, p in model.named_parameters()},
                     [None if m is None else m.float().cpu().clone() for m in tr.mems], tr.log_window())
    le, lg = out["eager"][0], out["graph"][0]
    assert all(abs(a - b) < 2e-4 for a, b in zip(le, lg)), (le, lg)
    for n in out["eager"][1]:
        a, b = out["eager"][1][n], out["graph"][1][n]
        assert float((a - b).abs().max()) <= 2e-5 + 2e-3 * float(a.abs().max()), (n, float((a - b).abs().max()))
    for a, b in zip(out["eager"][2], out["graph"][2]):
        assert (a is None) == (b is None)
        if a is not None:
            assert float((a - b).abs().max()) <= 2e-2 * float(a.abs().max())
    assert abs(out["eager"][3][0] - out["graph"][3][0]) < 1e-4 and out["eager"][3][2] == out["graph"][3][2]


@pytest.mark.parametrize("mem_len", [0, 64])
def test_merged_micro_batches_match_the_reference_loop(mem_len):
    """Trainer(merge_chunks=True) runs the `batch_chunk` micro-batches of a step (train.py:113-155) as ONE forward /
    backward over all columns whose loss weights every token by 1 / (batch_chunk x non-pad count of ITS micro-batch).
    Against the reference's loop (merge_chunks=False) on batches with a different number of pads in every micro-batch and
    per-column memory resets, dropout off: the gradient of the first step (cosine, max error), the loss of four
    consecutive steps (the XL memory carried as one tensor instead of one per micro-batch), the logging window; and the
    loss against the oracle's loop (oracle.xl_ref.train_step with batch_chunk = 4)."""
    from commu_amd.model.config_helper import get_cfg
    from commu_amd.model.dataset import BaseVocab
    from commu_amd.train import Trainer, build_model
    from oracle import xl_ref as X
    dev = torch.device(DEV)
    T, B, chunk = 64, 16, 4
    cfg = get_cfg(num_layers=2, num_heads=4, units=256, inner_size=512, tgt_length=T, mem_length=mem_len, batch_size=B,
                  batch_chunk=chunk, dropout=0.0, attention_dropout=0.0)
    g = torch.Generator().manual_seed(77)
    batches = []
    for i in range(4):
        stream = torch.randint(2, 729, (T + 1, B), generator=g)
        data, target = stream[:-1].clone(), stream[1:].clone()
        for c in range(chunk):                                  # pads: a different count in every micro-batch
            cols = slice(c * (B // chunk), (c + 1) * (B // chunk))
            npad = 5 + 11 * c + i
            target[T - npad:, cols.start] = 0
            data[T - npad + 1:, cols.start] = 0
        reset = torch.rand(B, generator=g) < (0.3 if mem_len else 0.0)
        batches.append((data.to(dev), target.to(dev), reset.to(dev), int((target != 0).sum())))
    runs = {}
    for merge in (True, False):
        model = build_model(cfg, BaseVocab(), dev, seed=9)
        model.train()
        tr = Trainer(model, cfg, num_gpus=1, merge_chunks=merge)
        d, t, r, n = batches[0]
        total, _, _ = tr._device_forward_backward(d, t, r, list(tr.mems) if tr.groups else [None] * (1 if merge else chunk))
        torch.cuda.synchronize()
        grad = model._ensure_flat()["g"].detach().float().cpu().clone()
        model.zero_grad()
        assert tr.groups == (chunk if merge else 1) and len(tr.mems) == (1 if merge else chunk)
        tr.mems = [None for _ in tr.mems]
        losses = [float(tr.step(*b)) for b in batches]
        runs[merge] = (float(total), grad, losses, tr.log_window())
    (tm, gm, lm, wm), (tl, gl, ll, wl) = runs[True], runs[False]
    assert abs(tm - tl) < 1e-3 * abs(tl), (tm, tl)
    cos = float((gm.double() * gl.double()).sum() / (gm.double().norm() * gl.double().norm()))
    err = float((gm - gl).abs().max()) / float(gl.abs().max())
    print(f"merged vs loop (mem_len {mem_len}): loss {tm:.5f} / {tl:.5f}, gradient cosine {cos:.6f}, max error {err:.2e}; "
          f"losses {lm} / {ll}")
    assert 0.9995 < cos < 1.0 + 1e-9 and err < 2e-2, (cos, err)
    assert all(abs(a - b) < 3e-3 * abs(b) for a, b in zip(lm, ll)), (lm, ll)
    assert abs(wm[0] - wl[0]) < 2e-3 * abs(wl[0]) and wm[2] == wl[2]
    # the oracle's loop over micro-batches on the first batch
    p = {k: v.detach().float().cpu().clone() for k, v in build_model(cfg, BaseVocab(), dev, seed=9).state_dict().items()
         if k not in ("crit.out_layers.0.weight", "pos_emb.inv_freq")}
    s = X.XLShape(2, 4, 256, 512)
    d, t, r, n = batches[0]
    oloss, _, _, _ = X.train_step(p, X.adam_init(p), s, d.cpu(), t.cpu(), r.cpu(), [None] * chunk, batch_chunk=chunk,
                                  mem_len=mem_len, same_length=False, lr_now=cfg.TRAIN.lr, clip=cfg.TRAIN.clip)
    assert abs(tm - float(oloss)) < 2e-2, (tm, float(oloss))
