"""Pins the CPU oracle (oracle/) against fixtures produced by the reference itself
(tests/golden/make_golden.py).  fp32 values: <= 2e-5 abs (same math, different op order);
integers / token ids: exact."""
import math
import os

import numpy as np
import pytest
import torch

from oracle import xl_ref as X
from oracle import decode_ref as Dz

TOL = 2e-5


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def params_of(z, prefix="p::"):
    return {k[len(prefix):]: torch.from_numpy(z[k]).clone() for k in z.files
            if k.startswith(prefix) and not k.endswith("pos_emb.inv_freq")}


def shape_of(meta):
    L, H, D, DI = [int(x) for x in meta[:4]]
    return X.XLShape(L, H, D, DI)


@pytest.mark.parametrize("tag", ["mem", "nomem", "dh50", "clamp"])      # clamp: cfg.MODEL.clamp_len = 9 (model.py:581-582)
def test_g1_forward_backward(golden_dir, tag):
    z = load(golden_dir, f"g1_train_{tag}.npz")
    s = shape_of(z["meta"])
    mem_len, same_length = int(z["meta"][6]), bool(z["meta"][7])
    clamp_len = int(z["clamp_len"]) if "clamp_len" in z.files else -1
    p = {k: v.requires_grad_(True) for k, v in params_of(z).items()}
    mems = None
    for seg in range(3):
        data, target = torch.from_numpy(z[f"data{seg}"]), torch.from_numpy(z[f"target{seg}"])
        reset = torch.from_numpy(z[f"reset{seg}"])
        nll, mems = X.forward_loss(p, s, data, target, reset, mems, mem_len, same_length, clamp_len=clamp_len)
        assert np.abs(nll.detach().numpy() - z[f"loss{seg}"]).max() < TOL
        if mem_len > 0:
            assert mems.shape == z[f"mems{seg}"].shape
            assert np.abs(mems.numpy() - z[f"mems{seg}"]).max() < TOL
        else:
            assert mems is None
        scalar = X.masked_mean_loss(nll, target)
        assert abs(float(scalar) - float(z[f"scalar{seg}"])) < TOL
    grads = torch.autograd.grad(scalar, list(p.values()))
    for (k, _), g in zip(p.items(), grads):
        ref = z["g::" + k]
        assert np.abs(g.numpy() - ref).max() < TOL + 1e-4 * np.abs(ref).max(), k


def test_g2_forward_generate(golden_dir):
    z = load(golden_dir, "g2_generate.npz")
    s = shape_of(z["meta"])
    p = params_of(z)
    with torch.no_grad():
        logits, mems = X.forward_generate(p, s, torch.from_numpy(z["ctx"]), None, 4146, True)
        assert np.abs(logits.numpy() - z["ctx_logits"]).max() < TOL
        assert np.abs(mems.numpy() - z["ctx_mems"]).max() < TOL
        for i, t in enumerate(z["toks"]):
            logits, mems = X.forward_generate(p, s, torch.tensor([[int(t)]]), mems, 4146, True)
            assert np.abs(logits.numpy() - z[f"step{i}_logits"]).max() < TOL
        assert np.abs(mems.numpy() - z["final_mems"]).max() < TOL
        mems = None
        for i in range(3):    # same_length with a short memory (mem_len 6, qlen 4)
            logits, mems = X.forward_generate(p, s, torch.from_numpy(z[f"sl_data{i}"]), mems, 6, True)
            assert np.abs(logits.numpy() - z[f"sl_logits{i}"]).max() < TOL
            assert np.abs(mems.numpy() - z[f"sl_mems{i}"]).max() < TOL


def test_g3_attention_layer(golden_dir):
    z = load(golden_dir, "g3_attn.npz")
    for c in range(int(z["ncase"])):
        pre = f"c{c}_"
        H, D, T, M, B, same_length, mem_len, reset_col = [int(x) for x in z[pre + "meta"]]
        s = X.XLShape(1, H, D, D)
        p = {"layers.0.dec_attn." + k[len(pre) + 3:]: torch.from_numpy(z[k]).clone().requires_grad_(True)
             for k in z.files if k.startswith(pre + "p::")}
        p["r_w_bias"] = torch.from_numpy(z[pre + "u"]).clone().requires_grad_(True)
        p["r_r_bias"] = torch.from_numpy(z[pre + "v"]).clone().requires_grad_(True)
        w = torch.from_numpy(z[pre + "w"]).clone().requires_grad_(True)
        mem = torch.from_numpy(z[pre + "mem"]) if M > 0 else None
        reset = None
        if reset_col >= 0:
            reset = torch.zeros(B, dtype=torch.bool)
            reset[reset_col] = True
        mask = X.attn_mask(T, M, B, reset, bool(same_length), mem_len)
        assert (mask.numpy() == z[pre + "mask"]).all()
        y = X.attn_block(p, 0, s, w, mem, X.sinusoid_table(T + M, D), mask)
        assert np.abs(y.detach().numpy() - z[pre + "y"]).max() < TOL
        y.backward(torch.from_numpy(z[pre + "gy"]))
        assert np.abs(w.grad.numpy() - z[pre + "gw"]).max() < 5e-5
        assert np.abs(p["r_w_bias"].grad.numpy() - z[pre + "gu"]).max() < 5e-5
        assert np.abs(p["r_r_bias"].grad.numpy() - z[pre + "gv"]).max() < 5e-5
        for k in z.files:
            if k.startswith(pre + "g::"):
                name = "layers.0.dec_attn." + k[len(pre) + 3:]
                assert np.abs(p[name].grad.numpy() - z[k]).max() < 1e-4, k


def test_g4_rel_shift_known_answer(golden_dir):
    z = load(golden_dir, "g45_tables.npz")
    ref = z["relshift_T5_M3"][0, 0]                     # x = arange(40).view(5, 8), T=5, M=3
    x = np.arange(40.0).reshape(5, 8)
    for i in range(5):
        for j in range(8):
            if j <= i + 3:                              # un-masked region only
                assert ref[i, j] == x[i, j + 5 - 1 - i]
    # oracle: BD[i, j] = (q_i + v) . r[j + T - 1 - i]; use one-hot rows so the score IS the index
    T, K = 5, 8
    q = torch.zeros(T, 1, 1, T * K)
    for i in range(T):
        q[i, 0, 0, i * K:(i + 1) * K] = 1.0            # picks r[m][i*K + m'] ...
    r = torch.zeros(K, 1, T * K)
    for m in range(K):
        for i in range(T):
            r[m, 0, i * K + m] = float(i * K + m)        # so (q_i . r[m]) = x[i, m]
    S = X.rel_attention_scores(q, torch.zeros(K, 1, 1, T * K), r, torch.zeros(1, T * K), torch.zeros(1, T * K))
    for i in range(T):
        for j in range(K):
            if j <= i + 3:
                assert float(S[0, 0, i, j]) == ref[i, j]
    assert (z["relshift_T1"][0, 0, 0] == np.arange(9.0)).all()


def test_g5_mask_tables(golden_dir):
    z = load(golden_dir, "g45_tables.npz")
    m = X.attn_mask(4, 2, 2, torch.tensor([False, True]), False, 8)
    assert (m.numpy() == z["mask_T4_M2"]).all()
    assert (X.attn_mask(4, 4, 1, None, True, 4).numpy() == z["mask_sl_T4_M4_ml4"]).all()
    assert (X.attn_mask(4, 2, 1, None, True, 8).numpy() == z["mask_sl_T4_M2_ml8"]).all()
    assert (X.attn_mask(3, 5, 2, torch.tensor([True, False]), True, 6).numpy() == z["mask_sl_T3_M5_ml6"]).all()
    sl = z["mask_sl_T4_M4_ml4"][0]
    assert ((~sl).sum(1) == 4).all()                    # every query sees exactly mem_len keys


@pytest.mark.parametrize("tag", ["greedy8", "sample8", "sample4x", "greedy5"])
def test_g6_decode_loop(golden_dir, tag):
    z = load(golden_dir, "g6_decode.npz")
    s = shape_of(z["meta"])
    p = params_of(z)
    p["crit.out_layers.0.bias"] = torch.from_numpy(z[f"{tag}_bias"]).clone()
    temp, nm, top_k, glen = z[f"{tag}_cfg"]
    meta = [int(t) for t in z["encoded_meta"]]
    calls = []

    def step(tok, mems):
        mlen_in = 0 if mems is None else mems.shape[1]
        with torch.no_grad():
            lg, nm_ = X.forward_generate(p, s, torch.tensor([[int(tok)]]), mems, 4146, True)
        calls.append((int(tok), mlen_in, nm_.shape[1]))
        return lg[-1, 0], nm_

    with torch.no_grad():                               # midi_inferrer.py:186-197 (Q3)
        _, mems = X.forward_generate(p, s, torch.tensor([0] + meta[:10])[:, None], None, 4146, True)
    seq = [0] + meta
    out = Dz.generate_sequence(step, seq, mems, chord_token=[int(t) for t in z[f"{tag}_chord_token"]],
                               chord_position=[int(t) for t in z[f"{tag}_chord_position"]],
                               num_measures=float(nm), temperature=float(temp), top_k=int(top_k),
                               uniforms=list(z[f"{tag}_uniforms"]), max_iters=int(glen))
    assert out == [int(t) for t in z[f"{tag}_seq"]]
    assert np.array_equal(np.array(calls), z[f"{tag}_trace"])


def test_g6_quirks_q3_q4(golden_dir):
    """The fixture itself shows Q3 (first fed token's memory discarded) and Q4 (last forced
    token fed twice): SURVEY.md section 0."""
    z = load(golden_dir, "g6_decode.npz")
    tr = z["sample8_trace"]
    assert tuple(tr[0]) == (727, 11, 12) and tr[1][1] == 11          # Q3
    fed = tr[:, 0].tolist()
    i = fed.index(199)
    assert fed[i + 1] == 199                                          # Q4


def test_g7_meta_known_answer(golden_dir):
    z = load(golden_dir, "g7_meta.npz")
    assert z["encoded_meta"].tolist() == [574, 623, 627, 635, 639, 642, 651, 684, 694, 720, 727]
    assert z["chord_token"].tolist() == [199, 285, 267, 258, 199, 285, 267, 258]
    assert z["chord_position"].tolist() == [432] * 8


def test_g8_lr_lambda(golden_dir):
    z = load(golden_dir, "g8_optim.npz")
    got = np.array([X.lr_lambda(s) for s in range(301)])
    assert np.array_equal(got, z["lr_lambda_0_300"])
    assert got[0] == 0.0                                # Q7
    far = np.array([X.lr_lambda(s) for s in (1000, 10000, 20000, 200000)])
    assert np.array_equal(far, z["lr_lambda_far"])


def test_g8_optimizer_steps(golden_dir):
    z = load(golden_dir, "g8_optim.npz")
    L, H, D, DI, T, B, mem_len, sl, chunk = [int(x) for x in z["meta"]]
    s = X.XLShape(L, H, D, DI)
    p = params_of(z)
    st = X.adam_init(p)
    mems = [None] * chunk
    warm = int(z["warmup"])
    for step in range(int(z["nsteps"])):
        lr_now = 0.004 * X.lr_lambda(step, warm, 0.004, 0.0001)
        assert abs(lr_now - float(z[f"lr{step}"])) < 1e-12
        loss, gn, mems, _ = X.train_step(
            p, st, s, torch.from_numpy(z[f"data{step}"]), torch.from_numpy(z[f"target{step}"]),
            torch.from_numpy(z[f"reset{step}"]), mems, batch_chunk=chunk, mem_len=mem_len,
            same_length=bool(sl), lr_now=lr_now, clip=float(z["clip"]))
        assert abs(loss - float(z[f"loss{step}"])) < 1e-5
        assert abs(gn - float(z[f"gnorm{step}"])) < 1e-4 * max(1.0, gn)
    for k, v in p.items():
        ref = z["after::" + k]
        assert np.abs(v.numpy() - ref).max() < 2e-5, k


def test_g9_sampling_probs(golden_dir):
    z = load(golden_dir, "g9_sampling.npz")
    for c in range(int(z["ncase"])):
        logits = torch.from_numpy(z[f"c{c}_logits"]).clone()
        temp = float(z[f"c{c}_temp"])
        view = logits[1:]
        for r in range(int(z[f"c{c}_rounds"])):
            probs = Dz.apply_sampling(Dz.calc_probs(view, temp), 32, z[f"c{c}_r{r}_wrong"].tolist())
            ref = z[f"c{c}_r{r}_probs"]
            assert np.abs(probs.numpy() - ref).max() < 1e-6
            assert ((probs.numpy() > 0) == (ref > 0)).all()
            assert probs[0] == 0                        # pad column, Q6
