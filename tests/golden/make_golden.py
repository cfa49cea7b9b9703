#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE.

Runs only in the build container (needs /root/reference, CPU only).  The fixtures are
data: seeded inputs and the reference's outputs.  Nothing of the reference's source is
stored.  Re-run with:  python tests/golden/make_golden.py

Third-party modules the reference imports but that are not installed here (yacs,
miditoolkit, parmap, pretty_midi) are replaced by in-memory stand-ins that are never
exercised on the model / sampling path.

Fixtures:
  g1_train_*.npz   model forward+backward over 3 consecutive segments (loss, mems, grads)
  g2_generate.npz  forward_generate: 11-token context then single-token steps
  g3_attn_*.npz    attention layer alone, output and input grads
  g45_tables.npz   _rel_shift known answer + mask tables
  g6_decode_*.npz  chord-forced decode loop traces (greedy, and sampled with injected variates)
  g8_optim.npz     lr_lambda table + optimiser steps (clip + Adam + LambdaLR)
  g8_dataset.npz   packed train iterator / rank-sharded eval iterator on a ragged corpus
  g9_sampling.npz  calc_probs + apply_sampling incl. compounded temperature
"""
import os
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def _install_stubs():
    class CfgNode(dict):
        def __getattr__(self, k):
            try:
                return self[k]
            except KeyError:
                raise AttributeError(k)

        def __setattr__(self, k, v):
            self[k] = v

        def freeze(self):
            pass

        def defrost(self):
            pass

    yacs = types.ModuleType("yacs")
    yacs_config = types.ModuleType("yacs.config")
    yacs_config.CfgNode = CfgNode
    yacs.config = yacs_config
    sys.modules["yacs"] = yacs
    sys.modules["yacs.config"] = yacs_config
    for name in ("miditoolkit", "parmap", "pretty_midi"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["miditoolkit"].MidiFile = object
    sys.path.insert(0, REF)


_install_stubs()

import torch  # noqa: E402
import torch.nn as nn  # noqa: E402

torch.set_num_threads(4)

from commu.model.model import MemTransformerLM  # noqa: E402


def ns(**kw):
    return types.SimpleNamespace(**kw)


def make_cfg(L, H, D, DI, tgt, mem, same_length=False, clamp_len=-1):
    return ns(MODEL=ns(num_layers=L, num_heads=H, units=D, inner_size=DI, dropout=0.0,
                       attention_dropout=0.0, same_length=same_length, clamp_len=clamp_len),
              TRAIN=ns(tgt_length=tgt, mem_length=mem),
              INITIALIZER=ns(base_init=0.01, embed_init=0.01))


class Vocab:
    pad_id = 0

    def __len__(self):
        return 729


def ref_train_py_snippet(first, last, env):
    """exec lines [first, last] (1-based, inclusive) of the reference train.py in ``env``.
    train.py cannot be imported (module-level CUDA/NCCL), so its pure-Python helpers are
    executed from where they lie."""
    lines = open(os.path.join(REF, "train.py")).read().split("\n")[first - 1:last]
    exec("\n".join(lines), env)
    return env


def build_model(cfg, seed, std=None):
    torch.manual_seed(seed)
    model = MemTransformerLM(cfg, Vocab())
    c = cfg
    if std is not None:
        c = make_cfg(1, 1, 1, 1, 1, 1)
        c.INITIALIZER = ns(base_init=std, embed_init=std)
    env = {"nn": nn, "cfg": c}
    ref_train_py_snippet(291, 342, env)            # init_weight .. weights_init
    model.apply(env["weights_init"])
    model.word_emb.apply(env["weights_init"])
    model.eval()
    return model


def sd_np(model):
    return {"p::" + k: v.detach().numpy().copy() for k, v in model.state_dict().items()
            if k != "crit.out_layers.0.weight"}


def save(name, **arrs):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrs)
    print(f"wrote {name}: {os.path.getsize(path) / 1024:.0f} KiB")


# ------------------------------------------------------------------------------------------------
def g1(tag, L, H, D, DI, T, B, mem_len, same_length, clamp_len=-1):
    cfg = make_cfg(L, H, D, DI, T, mem_len, same_length, clamp_len)
    model = build_model(cfg, 11, std=0.05)
    g = torch.Generator().manual_seed(5)
    out = sd_np(model)
    out["meta"] = np.array([L, H, D, DI, T, B, mem_len, int(same_length)])
    out["clamp_len"] = np.array(clamp_len)          # cfg.MODEL.clamp_len (model.py:581-582: positions above it share one row)
    mems = None
    for seg in range(3):
        data = torch.randint(1, 729, (T, B), generator=g)
        target = torch.randint(1, 729, (T, B), generator=g)
        if seg == 1:                    # a column that ends early: pad-filled tail
            data[T // 2:, 0] = 0
            target[T // 2:, 0] = 0
        reset = torch.zeros(B, dtype=torch.bool)
        if seg == 2:
            reset[1] = True
        model.zero_grad()
        loss, new_mems = model(data, target, reset, mems)
        scalar = loss[target != 0].float().mean()
        scalar.backward()
        out[f"data{seg}"] = data.numpy()
        out[f"target{seg}"] = target.numpy()
        out[f"reset{seg}"] = reset.numpy()
        out[f"loss{seg}"] = loss.detach().numpy()
        out[f"scalar{seg}"] = scalar.detach().numpy()
        if new_mems is not None:
            out[f"mems{seg}"] = new_mems.numpy()
        if seg == 2:
            for k, p in model.named_parameters():
                if k != "crit.out_layers.0.weight":
                    out["g::" + k] = p.grad.numpy().copy()
        mems = new_mems
    save(f"g1_train_{tag}.npz", **out)


def g2():
    cfg = make_cfg(2, 2, 64, 128, 128, 1024, True)
    model = build_model(cfg, 12, std=0.08)
    model.reset_length(1, 4146)
    out = sd_np(model)
    out["meta"] = np.array([2, 2, 64, 128, 1, 1, 4146, 1])
    ctx = torch.tensor([0, 574, 623, 627, 635, 639, 642, 651, 684, 694, 720])[:, None]
    with torch.no_grad():
        logits, mems = model.forward_generate(ctx, None)
        out["ctx"] = ctx.numpy()
        out["ctx_logits"] = logits.numpy()
        out["ctx_mems"] = mems.numpy()
        toks = [727, 2, 432, 199]
        for i, t in enumerate(toks):
            logits, mems = model.forward_generate(torch.tensor([[t]]), mems)
            out[f"step{i}_logits"] = logits.numpy()
        out["toks"] = np.array(toks)
        out["final_mems"] = mems.numpy()
        # batched variant (B=3) with a longer context, same_length mask exercised with short mem_len
        model.reset_length(4, 6)
        g = torch.Generator().manual_seed(3)
        mems = None
        for i in range(3):
            d = torch.randint(1, 729, (4, 3), generator=g)
            logits, mems = model.forward_generate(d, mems)
            out[f"sl_data{i}"] = d.numpy()
            out[f"sl_logits{i}"] = logits.numpy()
            out[f"sl_mems{i}"] = mems.numpy()
    save("g2_generate.npz", **out)


def g3():
    from commu.model.model import RelPartialLearnableMultiHeadAttn, PositionalEmbedding
    out = {}
    case = 0
    for (H, D, T, M, B, same_length, mem_len, reset_col) in [
            (2, 64, 8, 0, 2, False, 0, None), (2, 64, 8, 5, 3, False, 16, 1),
            (2, 64, 1, 12, 2, True, 64, None), (2, 64, 16, 16, 2, True, 16, None),
            (2, 100, 8, 5, 2, False, 16, None)]:
        dh = D // H
        torch.manual_seed(100 + case)
        att = RelPartialLearnableMultiHeadAttn(H, D, dh, 0.0, dropatt=0.0)
        for p in att.parameters():
            nn.init.normal_(p, 0.0, 0.08)
        nn.init.normal_(att.layer_norm.weight, 1.0, 0.08)
        att.eval()
        u = torch.randn(H, dh) * 0.1
        v = torch.randn(H, dh) * 0.1
        u.requires_grad_(True)
        v.requires_grad_(True)
        w = torch.randn(T, B, D, requires_grad=True)
        mem = torch.randn(M, B, D) if M > 0 else None
        K = T + M
        pos = PositionalEmbedding(D)(torch.arange(K - 1, -1, -1.0))
        # mask as MemTransformerLM._forward builds it (model.py:549-574)
        ones = torch.ones(T, K)
        if same_length:
            mask_len = K - mem_len
            msl = T - mask_len if mask_len > 0 else T
            mask = (torch.triu(ones, 1 + M) + torch.tril(ones, -msl)).bool().repeat(B, 1, 1)
        else:
            mask = torch.triu(ones, 1 + M).bool().repeat(B, 1, 1)
        if reset_col is not None:
            mask[reset_col, :, :M] = True
        y = att(w, pos, u, v, attn_mask=mask, mems=mem)
        gy = torch.randn_like(y)
        y.backward(gy)
        pre = f"c{case}_"
        out[pre + "meta"] = np.array([H, D, T, M, B, int(same_length), mem_len,
                                      -1 if reset_col is None else reset_col])
        for k, p in att.named_parameters():
            out[pre + "p::" + k] = p.detach().numpy().copy()
            out[pre + "g::" + k] = p.grad.numpy().copy()
        out[pre + "u"], out[pre + "v"] = u.detach().numpy(), v.detach().numpy()
        out[pre + "gu"], out[pre + "gv"] = u.grad.numpy(), v.grad.numpy()
        out[pre + "w"], out[pre + "gw"] = w.detach().numpy(), w.grad.numpy()
        if mem is not None:
            out[pre + "mem"] = mem.numpy()
        out[pre + "mask"] = mask.numpy()
        out[pre + "y"], out[pre + "gy"] = y.detach().numpy(), gy.numpy()
        case += 1
    out["ncase"] = np.array(case)
    save("g3_attn.npz", **out)


def g45():
    from commu.model.model import RelMultiHeadAttn
    att = RelMultiHeadAttn(1, 8, 8, 0.0)
    x = torch.arange(40.0).view(1, 1, 5, 8)
    out = {"relshift_T5_M3": att._rel_shift(x).numpy()}
    x1 = torch.arange(9.0).view(1, 1, 1, 9)
    out["relshift_T1"] = att._rel_shift(x1).numpy()

    def model_mask(T, M, B, same_length, mem_len, reset):
        cfg = make_cfg(1, 1, 8, 8, T, mem_len, same_length)
        m = build_model(cfg, 1)
        captured = {}

        def fake_layer(core, pos, u, v, dec_attn_mask=None, mems=None):
            captured["mask"] = dec_attn_mask.clone()
            return core
        m.layers[0].forward = fake_layer
        mems = torch.zeros(2, M, B, 8)
        with torch.no_grad():
            m._forward(torch.ones(T, B, dtype=torch.long), reset, mems=mems)
        return captured["mask"].numpy()

    out["mask_T4_M2"] = model_mask(4, 2, 2, False, 8, torch.tensor([False, True]))
    out["mask_sl_T4_M4_ml4"] = model_mask(4, 4, 1, True, 4, None)
    out["mask_sl_T4_M2_ml8"] = model_mask(4, 2, 1, True, 8, None)
    out["mask_sl_T3_M5_ml6"] = model_mask(3, 5, 2, True, 6, torch.tensor([True, False]))
    save("g45_tables.npz", **out)


# ------------------------------------------------------------------------------------------------
ENCODED_META = [574, 623, 627, 635, 639, 642, 651, 684, 694, 720, 727]   # README example (G7)
CHORD_TOKEN = [199, 285, 267, 258, 199, 285, 267, 258]
CHORD_POSITION = [432] * 8


def g6():
    from commu.midi_generator import midi_inferrer as mi
    import commu.midi_generator.midi_inferrer  # noqa: F401

    cfg = make_cfg(2, 2, 64, 128, 128, 1024, True)
    model = build_model(cfg, 24, std=0.12)
    model.reset_length(1, 4146)
    out = sd_np(model)
    out["meta"] = np.array([2, 2, 64, 128, 1, 1, 4146, 1])
    out["encoded_meta"] = np.array(ENCODED_META)

    def bias_vec(bar, pos, chord, eos, meta):
        # output bias shaping so that a random-weight model emits REMI-like structure
        b = torch.zeros(729)
        b[2], b[1] = bar, eos
        b[432:560] = pos
        b[195:304] = chord
        b[560:] = meta
        return b
    greedy_bias = bias_vec(3.0, 1.0, -4.0, 1.0, -2.0)
    sample_bias = bias_vec(3.0, 1.0, 1.0, 1.0, -2.0)

    calls = []
    orig_fg = model.forward_generate

    def traced(data, mems):
        mlen_in = 0 if mems is None else mems.shape[1]
        logits, new = orig_fg(data, mems)
        calls.append((data.flatten().tolist(), mlen_in, new.shape[1],
                      logits[-1, 0].detach().clone()))
        return logits, new
    model.forward_generate = traced

    final = {}

    def no_validate(self, seq):
        final["seq"] = list(seq)
        final["remnant"] = len(self.chord_token)
    mi.TeacherForceTask.validate_teacher_forced_sequence = no_validate

    cases = [
        ("greedy8", 0.0, 8, list(CHORD_TOKEN), list(CHORD_POSITION), 260, None, greedy_bias),
        ("sample8", 0.95, 8, list(CHORD_TOKEN), list(CHORD_POSITION), 400, 7, sample_bias),
        # multi-chord-per-bar progression on 4 measures: intra-bar chord positions
        ("sample4x", 0.95, 4, [199, 285, 267, 258, 199, 285], [432, 496, 432, 432, 496, 432], 400, 9,
         sample_bias),
        ("greedy5", 0.0, 5, [199, 285, 267, 258], [432] * 4, 200, None, greedy_bias),
        # sampled, every variate at least MARGIN away from every step of the CDF it is drawn from: reduced-precision
        # logits cannot flip a draw, so this trace must be reproduced token for token
        ("sample8m", 0.95, 8, list(CHORD_TOKEN), list(CHORD_POSITION), 400, 11, sample_bias),
    ]
    MARGIN = 0.04
    for tag, temp, nm, ctok, cpos, glen, useed, bias in cases:
        calls.clear()
        final.clear()
        model.crit.out_layers[0].bias.data.copy_(bias)
        out[f"{tag}_bias"] = bias.numpy().copy()
        input_data = ns(temperature=temp, top_k=32, num_generate=1, num_measures=nm,
                        chord_token_components={"chord_token": list(ctok),
                                                "chord_position": list(cpos)})
        task = mi.InferenceTask(torch.device("cpu"))
        task(model=model, input_data=input_data,
             inference_cfg=ns(GENERATION=ns(generation_length=glen)))
        uniforms, draw_tokens, draw_margin = [], [], []
        if useed is not None:
            rng = np.random.RandomState(useed)
            want_margin = MARGIN if tag.endswith("m") else 0.0

            def infer(self, probs, rng=rng, uniforms=uniforms, want_margin=want_margin):
                cdf = torch.cumsum(probs.double(), 0)
                best = None
                for _ in range(400):
                    u = float(rng.uniform(0.02, 0.98))
                    margin = float((cdf - u).abs().min())
                    if best is None or margin > best[1]:
                        best = (u, margin)
                    if margin >= want_margin:
                        break
                u, margin = best
                uniforms.append(u)
                t = int(torch.searchsorted(cdf, torch.tensor(u, dtype=torch.float64), right=True))
                t = min(t, probs.numel() - 1)
                draw_tokens.append(t)
                draw_margin.append(margin)
                return t
            mi.InferenceTask.infer_token = infer
        with torch.no_grad():
            seq, mems = task.init_seq_and_mems(list(ENCODED_META), len(ENCODED_META))
            task.generate_sequence(seq, mems)
        out[f"{tag}_cfg"] = np.array([temp, nm, 32, glen], dtype=np.float64)
        out[f"{tag}_chord_token"] = np.array(ctok)
        out[f"{tag}_chord_position"] = np.array(cpos)
        out[f"{tag}_uniforms"] = np.array(uniforms, dtype=np.float64)
        out[f"{tag}_draw_tokens"] = np.array(draw_tokens, dtype=np.int64)          # token of every draw, rejected ones too
        out[f"{tag}_draw_margin"] = np.array(draw_margin, dtype=np.float64)        # |cdf - u| to the nearest CDF step
        out[f"{tag}_seq"] = np.array(final["seq"])
        out[f"{tag}_remnant"] = np.array(final["remnant"])
        out[f"{tag}_trace"] = np.array([[c[0][-1], c[1], c[2]] for c in calls[1:]])
        lg = torch.stack([c[3] for c in calls[1:]])
        top2 = torch.topk(lg[:, 1:], 2, dim=-1).values
        out[f"{tag}_min_gap"] = (top2[:, 0] - top2[:, 1]).min().numpy()
        out[f"{tag}_gaps"] = (top2[:, 0] - top2[:, 1]).numpy()
        out[f"{tag}_logits"] = lg[:48].numpy().astype(np.float32)
        print(tag, "len", len(final["seq"]), "calls", len(calls), "remnant", final["remnant"],
              "min gap", float(out[f"{tag}_min_gap"]))
    save("g6_decode.npz", **out)


def g8_optim():
    import torch.optim as optim
    out = {}
    env = {"cfg": ns(TRAIN=ns(warmup_step=100, lr=0.004, lr_min=0.0001))}
    ref_train_py_snippet(448, 460, env)
    out["lr_lambda_0_300"] = np.array([env["lr_lambda"](s) for s in range(0, 301)])
    out["lr_lambda_far"] = np.array([env["lr_lambda"](s) for s in (1000, 10000, 20000, 200000)])

    L, H, D, DI, T, B, mem_len, chunk = 2, 2, 64, 128, 12, 4, 16, 2
    cfg = make_cfg(L, H, D, DI, T, mem_len, False)
    model = build_model(cfg, 31, std=0.05)
    out.update(sd_np(model))
    out["meta"] = np.array([L, H, D, DI, T, B, mem_len, 0, chunk])
    lr = 0.004
    opt = optim.Adam(model.parameters(), lr=lr, weight_decay=0.0)
    env2 = {"cfg": ns(TRAIN=ns(warmup_step=2, lr=lr, lr_min=0.0001))}
    ref_train_py_snippet(448, 460, env2)
    sched = optim.lr_scheduler.LambdaLR(opt, lr_lambda=env2["lr_lambda"])
    g = torch.Generator().manual_seed(8)
    mems = [None] * chunk
    nsteps = 4
    for step in range(nsteps):
        data = torch.randint(1, 729, (T, B), generator=g)
        target = torch.randint(1, 729, (T, B), generator=g)
        reset = torch.zeros(B, dtype=torch.bool)
        if step == 2:
            reset[2] = True
            target[T - 3:, 1] = 0
            data[T - 3:, 1] = 0
        # train.py:133-169
        model.zero_grad()
        dch, tch, rch = torch.chunk(data, chunk, 1), torch.chunk(target, chunk, 1), torch.chunk(reset, chunk, 0)
        loss_sum = 0.0
        for i in range(chunk):
            loss, mems[i] = model(dch[i].contiguous(), tch[i].contiguous(), rch[i].contiguous(), mems[i])
            loss = loss[tch[i] != 0].float().mean() / chunk
            loss_sum += loss.item()
            loss.backward()
        gn = torch.nn.utils.clip_grad_norm_(model.parameters(), 0.25)
        out[f"lr{step}"] = np.array(opt.param_groups[0]["lr"])
        opt.step()
        opt.zero_grad()
        sched.step()
        out[f"data{step}"], out[f"target{step}"], out[f"reset{step}"] = data.numpy(), target.numpy(), reset.numpy()
        out[f"loss{step}"] = np.array(loss_sum)
        out[f"gnorm{step}"] = gn.detach().numpy()
    for k, v in model.state_dict().items():
        if k != "crit.out_layers.0.weight":
            out["after::" + k] = v.detach().numpy().copy()
    out["nsteps"] = np.array(nsteps)
    out["clip"] = np.array(0.25)
    out["warmup"] = np.array(2)
    save("g8_optim.npz", **out)


def g10():
    """A checkpoint exactly as the reference writes it (train.py:39-48: unwrapped state_dict, torch.optim.Adam state,
    LambdaLR state, the reference's own BaseVocab instance, amp None) after two optimiser steps, plus what the
    reference computes FROM it: generation logits of the model that model_initializer.py:36-51 builds from the file,
    the parameters after one more optimiser step resumed from its optimiser state, and per-tensor statistics of
    weights_init (train.py:291-342) on a wider model."""
    import torch.optim as optim
    from commu.model.dataset import BaseVocab as RefVocab
    L, H, D, DI, T, B, mem_len = 2, 2, 32, 64, 10, 4, 12
    cfg = make_cfg(L, H, D, DI, T, mem_len, False)
    model = build_model(cfg, 41, std=0.08)
    model.train()
    lr = 0.004
    opt = optim.Adam(model.parameters(), lr=lr, weight_decay=0.0)
    env = {"cfg": ns(TRAIN=ns(warmup_step=2, lr=lr, lr_min=0.0001))}
    ref_train_py_snippet(448, 460, env)
    sched = optim.lr_scheduler.LambdaLR(opt, lr_lambda=env["lr_lambda"])
    g = torch.Generator().manual_seed(12)

    def step(mems):
        data = torch.randint(1, 729, (T, B), generator=g)
        target = torch.randint(1, 729, (T, B), generator=g)
        model.zero_grad()
        loss, mems = model(data, target, torch.zeros(B, dtype=torch.bool), mems)      # train.py:139-155 (one chunk)
        loss[target != 0].float().mean().backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)                       # :159-161
        opt.step()                                                                    # :164
        sched.step()                                                                  # :169
        return data, target, mems
    mems = None
    for _ in range(3):
        _, _, mems = step(mems)
    checkpoint = {"model": model.state_dict(), "optimizer": opt.state_dict(), "train_step": 3,
                  "scheduler": sched.state_dict(), "best_val_loss": 6.25, "vocab": RefVocab()}
    checkpoint["amp"] = None
    path = os.path.join(HERE, "g10_checkpoint.pt")
    torch.save(checkpoint, path)
    print(f"wrote g10_checkpoint.pt: {os.path.getsize(path) / 1024:.0f} KiB")
    out = {"meta": np.array([L, H, D, DI, T, B, mem_len])}
    # --- what model_initializer.py:36-51 does with the file (same_length=True, strict=False, eval, reset_length)
    gen_cfg = make_cfg(L, H, D, DI, T, mem_len, True)
    gm = MemTransformerLM(gen_cfg, Vocab())
    gm.load_state_dict(torch.load(path, weights_only=False)["model"], strict=False)
    gm.eval()
    gm.reset_length(1, 4146)
    ctx = torch.tensor([0] + ENCODED_META[:10])[:, None]
    with torch.no_grad():
        lg, gmems = gm.forward_generate(ctx, None)
        out["gen_ctx"] = ctx.numpy()
        out["gen_logits0"] = lg.numpy()
        toks = [ENCODED_META[10], 2, 432]
        for i, t in enumerate(toks):
            lg, gmems = gm.forward_generate(torch.tensor([[t]]), gmems)
            out[f"gen_logits{i + 1}"] = lg.numpy()
        out["gen_tokens"] = np.array(toks)
    # --- resume: one more optimiser step from the saved optimiser / scheduler state (no memory carried over)
    data, target, _ = step(None)
    out["resume_data"], out["resume_target"] = data.numpy(), target.numpy()
    out["resume_lr"] = np.array(opt.param_groups[0]["lr"])
    for k, v in model.state_dict().items():
        if k != "crit.out_layers.0.weight":
            out["after::" + k] = v.detach().numpy().copy()
    # --- weights_init statistics (reference init on a model wide enough for stable moments)
    wcfg = make_cfg(2, 4, 256, 512, 16, 16, False)
    wm = build_model(wcfg, 5)
    for k, v in wm.state_dict().items():
        if k in ("crit.out_layers.0.weight", "pos_emb.inv_freq"):
            continue
        out["init::" + k] = np.array([float(v.mean()), float(v.std()) if v.numel() > 1 else 0.0, float(v.abs().max())])
    save("g10_checkpoint.npz", **out)


def g8_dataset():
    from commu.model.dataset import ComMUDataset
    rng = np.random.RandomState(4)

    def corpus(n):
        metas, events = [], []
        for _ in range(n):
            metas.append(np.array(rng.randint(560, 729, size=11), dtype=object))
            ln = int(rng.randint(5, 60))
            ev = rng.randint(2, 560, size=ln).astype(np.int16)
            ev[-1] = 1
            events.append(ev)
        return metas, events

    out = {}
    with tempfile.TemporaryDirectory() as d:
        for split, n in (("train", 23), ("val", 9)):
            metas, events = corpus(n)
            np.save(os.path.join(d, f"input_{split}.npy"), np.array(metas, dtype=object), allow_pickle=True)
            ev = np.empty(n, dtype=object)
            for i, e in enumerate(events):
                ev[i] = e
            np.save(os.path.join(d, f"target_{split}.npy"), ev, allow_pickle=True)
            out[f"{split}_meta"] = np.stack([m.astype(np.int64) for m in metas])
            out[f"{split}_events"] = np.concatenate(events).astype(np.int64)
            out[f"{split}_lens"] = np.array([len(e) for e in events])
        ds = ComMUDataset(d, None)
    B, T = 4, 16
    it = ds.get_iterator(B, T, "cpu", "train", True, seed=1111)()
    for i in range(14):                       # long enough to wrap the epoch once (reshuffle)
        data, target, reset, ntok = next(it)
        out[f"tr{i}_data"], out[f"tr{i}_target"] = data.numpy().copy(), target.numpy().copy()
        out[f"tr{i}_reset"], out[f"tr{i}_ntok"] = reset.numpy().copy(), np.array(ntok)
    out["tr_n"] = np.array(14)
    it = ds.get_iterator(3, 8, "cpu", "valid", False, seed=None)()
    nb = 0      # the reference re-uses its batch tensors (.to("cpu") is a no-op): copy while iterating
    for i, (data, target, reset, ntok) in enumerate(it):
        out[f"ns{i}_data"], out[f"ns{i}_target"] = data.numpy().copy(), target.numpy().copy()
        out[f"ns{i}_reset"], out[f"ns{i}_ntok"] = reset.numpy().copy(), np.array(ntok)
        nb += 1
    out["noshuf_n"] = np.array(nb)
    for ws in (1, 2):
        for rank in range(ws):
            nb = 0
            for i, (data, target, allreset, ntok) in enumerate(
                    ds.eval_iterator(4, 16, "cpu", "valid", local_rank=rank, world_size=ws)()):
                pre = f"ev_ws{ws}_r{rank}_{i}_"
                out[pre + "data"], out[pre + "target"] = data.numpy().copy(), target.numpy().copy()
                out[pre + "reset"], out[pre + "ntok"] = np.array(allreset), np.array(ntok)
                nb += 1
            out[f"ev_ws{ws}_r{rank}_n"] = np.array(nb)
    save("g8_dataset.npz", **out)


def g9():
    from commu.midi_generator import midi_inferrer as mi
    out = {}
    g = torch.Generator().manual_seed(17)
    case = 0
    for temp, wrong_seq in [(0.95, [[]]), (0.95, [[], [199], [199, 285]]), (1.3, [[], [250]]),
                            (0.0, [[]]), (0.5, [[], [300], [300, 301], [300, 301, 302]])]:
        logits729 = torch.randn(729, generator=g) * 3.0
        logits729[195:304] += 2.0              # make chord tokens likely so masking matters
        task = mi.InferenceTask(torch.device("cpu"))
        task(model=None, input_data=ns(temperature=temp, top_k=32), inference_cfg=None)
        out[f"c{case}_logits"] = logits729.numpy().copy()
        out[f"c{case}_temp"] = np.array(temp)
        view = logits729[1:]                   # the view calc_logits_and_mems returns (Q5/Q6)
        for r, wrong in enumerate(wrong_seq):
            probs = task.calc_probs(view)
            probs = task.apply_sampling(probs, wrong)
            out[f"c{case}_r{r}_wrong"] = np.array(wrong, dtype=np.int64)
            out[f"c{case}_r{r}_probs"] = probs.numpy().copy()
        out[f"c{case}_rounds"] = np.array(len(wrong_seq))
        case += 1
    out["ncase"] = np.array(case)
    save("g9_sampling.npz", **out)


def g7():
    """Meta encoding known answer for the README example (needs pydantic v1-style API)."""
    try:
        from commu.midi_generator.info_preprocessor import PreprocessTask
        args = dict(output_dir="/tmp/x", bpm=70, audio_key="aminor", time_signature="4/4",
                    pitch_range="mid_high", num_measures=8.0, inst="acoustic_piano", genre="newage",
                    min_velocity=60, max_velocity=80, track_role="main_melody", rhythm="standard",
                    chord_progression="Am-Am-Am-Am-Am-Am-Am-Am-G-G-G-G-G-G-G-G-F-F-F-F-F-F-F-F-E-E-E-E-E-E-E-E-"
                                      "Am-Am-Am-Am-Am-Am-Am-Am-G-G-G-G-G-G-G-G-F-F-F-F-F-F-F-F-E-E-E-E-E-E-E-E",
                    num_generate=1, top_k=32, temperature=0.95)
        t = PreprocessTask()
        enc = t.execute(args)
        comp = t.input_data.chord_token_components
        save("g7_meta.npz", encoded_meta=np.array(enc), chord_token=np.array(comp["chord_token"]),
             chord_position=np.array(comp["chord_position"]))
        print("g7", enc, comp)
    except Exception as e:   # pragma: no cover
        print("g7 skipped:", repr(e))


def g7_cases():
    """Meta / chord-progression encoding over many seeded inputs (incl. "unknown" fields, 3/4 6/8 12/8 meters,
    flat / slash / extended chord names, rejected inputs): inputs and the reference's outputs or error class."""
    import json
    import random
    from commu.midi_generator.info_preprocessor import PreprocessTask
    from commu.preprocessor.utils import constants as C
    rng = random.Random(7)
    naturals, sharps, flats = list("ABCDEFG"), ["A#", "C#", "D#", "F#", "G#"], ["Ab", "Bb", "Db", "Eb", "Gb"]
    base_q = ["", "7", "+", "dim", "m", "m7", "m7b5", "maj7", "sus4"]
    nat_q = base_q + ["7sus4", "m6", "sus2", "add2", "dim7", "6", "madd2"]
    flat_q = base_q + ["maj", "dim7", "m6", "7sus4", "sus2", "add2", "6", "madd2"]
    bad_q = ["9", "mM7", "maj"]                        # not in the vocabulary for the root they are drawn with

    def one_chord():
        r = rng.random()
        if r < 0.45:
            c = rng.choice(naturals) + rng.choice(nat_q)
        elif r < 0.65:
            c = rng.choice(sharps) + rng.choice(base_q)
        elif r < 0.97:
            c = rng.choice(flats) + rng.choice(flat_q)
        else:
            c = rng.choice(sharps) + rng.choice(bad_q)
        if rng.random() < 0.15:
            c += "/" + rng.choice(naturals + sharps + flats)
        if rng.random() < 0.08:
            c += "(add9)"
        return c
    cases = []
    for n in range(260):
        ts = rng.choice(list(C.TIME_SIG_MAP))
        nm = rng.choice([4, 5, 8, 9, 16, 17, 4.0, 8.0, 8, 4, 16, 12, 3])
        from fractions import Fraction
        nchord = int((nm - (nm % 4)) * Fraction(ts) * 8) if nm >= 4 else 8
        prog, cur = [], None
        for i in range(max(nchord, 1)):
            if cur is None or rng.random() < 0.3:
                cur = one_chord()
            prog.append(cur)
        if rng.random() < 0.03:
            prog = prog[:-1]                       # wrong length -> validator error
        unk = lambda v: C.UNKNOWN if (isinstance(v, str) and rng.random() < 0.06) else v
        args = dict(output_dir="/tmp/x", bpm=unk(rng.choice([1, 3, 40, 70, 120, 199, 200, 260])),
                    audio_key=unk(rng.choice(list(C.KEY_MAP) + ["hminor"])),
                    time_signature=ts, pitch_range=unk(rng.choice(list(C.PITCH_RANGE_MAP))), num_measures=nm,
                    inst=unk(rng.choice(list(C.INST_MAP) + ["kazoo"])), genre=unk(rng.choice(list(C.GENRE_MAP))),
                    min_velocity=unk(rng.randrange(1, 127)), max_velocity=unk(rng.randrange(1, 128)),
                    track_role=unk(rng.choice(list(C.TRACK_ROLE_MAP))), rhythm=unk(rng.choice(list(C.RHYTHM_MAP))),
                    chord_progression="-".join(prog), num_generate=1, top_k=32, temperature=0.95)
        rec = {"args": {k: v for k, v in args.items() if k != "output_dir"}}
        try:
            t = PreprocessTask()
            rec["encoded_meta"] = [int(x) for x in t.execute(dict(args))]
        except Exception as e:
            rec["meta_error"] = type(e).__name__
        try:
            t2 = PreprocessTask()
            t2.normalize_input_data(dict(args))
            comp = t2.input_data.chord_token_components
            rec["chord_token"] = [int(x) for x in comp["chord_token"]]
            rec["chord_position"] = [int(x) for x in comp["chord_position"]]
        except Exception as e:
            rec["chord_error"] = type(e).__name__
        cases.append(rec)
    import gzip
    with gzip.open(os.path.join(HERE, "g7_meta_cases.json.gz"), "wt") as f:
        json.dump(cases, f)
    print("g7_cases", len(cases), sum("encoded_meta" in c for c in cases), sum("chord_token" in c for c in cases))


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g2", "g3", "g45", "g6", "g7", "g7c", "g8o", "g8d", "g9", "g10"]
    if "g1" in which:
        g1("mem", 2, 2, 64, 128, 12, 3, 16, False)
        g1("nomem", 2, 2, 64, 128, 12, 3, 0, False)
        g1("dh50", 2, 2, 100, 136, 10, 2, 12, False)
    if "g1" in which or "g1c" in which:
        g1("clamp", 2, 2, 64, 128, 12, 3, 16, False, clamp_len=9)          # 28 key positions, the 18 farthest share position 9
    if "g2" in which:
        g2()
    if "g3" in which:
        g3()
    if "g45" in which:
        g45()
    if "g6" in which:
        g6()
    if "g10" in which:
        g10()
    if "g7" in which:
        g7()
    if "g7c" in which:
        g7_cases()
    if "g8o" in which:
        g8_optim()
    if "g8d" in which:
        g8_dataset()
    if "g9" in which:
        g9()
