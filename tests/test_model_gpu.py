"""GPU parity of the whole model / training step against the golden fixtures produced by the
reference (tests/golden/g1_*, g2, g8_optim) and against the CPU oracle on fresh seeded inputs.

The product computes GEMM/attention operands in bf16 with fp32 accumulation (BASELINE.json
configs[1] "bf16 training"); tolerances are therefore bf16-level and stated at each assert.
"""
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import xl_ref as X  # noqa: E402

DEV = "cuda"


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def build_from_fixture(z, same_length=None):
    from commu_amd.model.config_helper import get_cfg
    from commu_amd.model.dataset import BaseVocab
    from commu_amd.model.model import MemTransformerLM
    L, H, D, DI, T, B, mem_len, sl = [int(x) for x in z["meta"][:8]]
    cfg = get_cfg(num_layers=L, num_heads=H, units=D, inner_size=DI, tgt_length=T, mem_length=mem_len,
                  dropout=0.0, attention_dropout=0.0, same_length=bool(sl) if same_length is None else same_length,
                  clamp_len=int(z["clamp_len"]) if "clamp_len" in z.files else -1)
    model = MemTransformerLM(cfg, BaseVocab())
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("p::")}
    sd["crit.out_layers.0.weight"] = sd["word_emb.emb_layers.0.weight"]
    missing, unexpected = model.load_state_dict(sd, strict=True), None
    return model.to(DEV), cfg


def _dump(tag, obj):
    import json
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, f"diag_{tag}.json"), "w") as f:
        json.dump(obj, f, indent=1, default=float)


def relerr(a, b):
    a = torch.as_tensor(a).detach().float().cpu()
    b = torch.as_tensor(b).detach().float().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


# dh50: d_model 100, d_head 50 (zero-padded to 128 / 64); clamp: cfg.MODEL.clamp_len = 9 (the reference's outputs with it)
@pytest.mark.parametrize("tag", ["mem", "nomem", "dh50", "clamp"])
def test_g1_forward_backward_vs_reference(golden_dir, tag):
    z = load(golden_dir, f"g1_train_{tag}.npz")
    model, cfg = build_from_fixture(z)
    model.eval()
    mems = None
    for seg in range(3):
        data = torch.from_numpy(z[f"data{seg}"]).to(DEV)
        target = torch.from_numpy(z[f"target{seg}"]).to(DEV)
        reset = torch.from_numpy(z[f"reset{seg}"]).to(DEV)
        model.zero_grad()
        loss, mems = model(data, target, reset, mems)
        ref = torch.from_numpy(z[f"loss{seg}"])
        err = (loss.detach().cpu() - ref).abs()
        # per-token NLL ~ 6.6; bf16 operands: <= 4e-2 abs worst token, <= 6e-3 on average
        assert float(err.max()) < 4e-2 and float(err.mean()) < 6e-3, (float(err.max()), float(err.mean()))
        if tag != "nomem":
            assert mems.shape == z[f"mems{seg}"].shape
            assert relerr(mems, z[f"mems{seg}"]) < 2e-2          # hidden states stored as bf16
        else:
            assert mems is None
        scalar = loss[target != 0].float().mean()
        assert abs(float(scalar) - float(z[f"scalar{seg}"])) < 5e-3
        scalar.backward()
    worst, cosines = {}, {}
    for name, p in model.named_parameters():
        ref = torch.from_numpy(z["g::" + name]).flatten()
        got = p.grad.detach().float().cpu().flatten()
        worst[name] = relerr(got, ref)
        cosines[name] = float(torch.dot(got, ref) / (got.norm() * ref.norm() + 1e-30))
    # Adam's FIRST update of a coordinate is -lr * sign(g) (m / sqrt(v) = g / |g|), so the cosine between the build's and the
    # reference's first updates of a tensor is the mean SIGN agreement of its gradient -- and a coordinate whose gradient lies
    # below the bf16 noise of the pass flips its sign half the time however small it is.  sign_cos: that predicted update cosine;
    # sign_mass: the share of the reference gradient's L1 mass on coordinates whose sign agrees.  (This is what the r_net.weight
    # window of test_g8 is about: sign_cos is far below 1 there while sign_mass shows that only negligible coordinates flip.)
    sign_cos, sign_mass = {}, {}
    for name, p in model.named_parameters():
        ref = torch.from_numpy(z["g::" + name]).flatten()
        got = p.grad.detach().float().cpu().flatten()
        agree = torch.sign(got) == torch.sign(ref)
        sign_cos[name] = float((torch.sign(got) * torch.sign(ref)).mean())
        sign_mass[name] = float((ref.abs() * agree).sum() / (ref.abs().sum() + 1e-30))
    _dump(f"g1_{tag}", {"relerr": worst, "cos": cosines, "sign_cos": sign_cos, "sign_mass": sign_mass})
    # every gradient tensor points the same way as the reference's ...
    assert min(cosines.values()) > 0.995, cosines
    # ... and whatever coordinates disagree in SIGN carry next to none of the gradient
    assert min(sign_mass.values()) > 0.995, sign_mass          # (measured >= 0.998; r_net.weight 0.9995-0.9998 at sign_cos 0.70-0.80)
    # ... and matches element-wise to bf16 accuracy (chain through 2 layers).  The first FFN Linear is not held to
    # an element-wise bound HERE: a pre-activation within bf16 rounding of 0 flips its ReLU gate against the fp32
    # reference, which changes one whole term of that unit's weight / bias gradient (36 tokens here) -- a property of
    # ReLU.  Its element-wise check is test_g1_gradients_with_the_builds_relu_gates_injected (6e-2, the build's own
    # gates given to the oracle); the cosine bound above covers it here.
    for k, v in worst.items():
        if "pos_ff.CoreNet.0" not in k:
            assert v < 6e-2, (k, v)


def test_g1_masked_mean_matches_indexing(golden_dir):
    from commu_amd.functional import masked_mean
    z = load(golden_dir, "g1_train_mem.npz")
    model, _ = build_from_fixture(z)
    data, target = torch.from_numpy(z["data1"]).to(DEV), torch.from_numpy(z["target1"]).to(DEV)
    loss, _ = model(data, target, torch.zeros(3, dtype=torch.bool, device=DEV), None)
    a = masked_mean(loss, target, 0, 0.5)
    b = loss[target != 0].float().mean() * 0.5
    assert abs(float(a) - float(b)) < 1e-5
    model.zero_grad()
    a.backward()
    g1 = {n: p.grad.clone() for n, p in model.named_parameters()}
    model.zero_grad()
    loss, _ = model(data, target, torch.zeros(3, dtype=torch.bool, device=DEV), None)
    (loss[target != 0].float().mean() * 0.5).backward()
    for n, p in model.named_parameters():
        assert relerr(g1[n], p.grad) < 1e-3, n


def test_g2_forward_generate_vs_reference(golden_dir):
    z = load(golden_dir, "g2_generate.npz")
    model, _ = build_from_fixture(z)
    model.eval()
    model.reset_length(1, 4146)
    logits, mems = model.forward_generate(torch.from_numpy(z["ctx"]).to(DEV), None)
    scale = float(np.abs(z["ctx_logits"]).max())
    assert relerr(logits, z["ctx_logits"]) < 2e-2, scale
    assert mems.shape == z["ctx_mems"].shape
    for i, t in enumerate(z["toks"]):
        logits, mems = model.forward_generate(torch.tensor([[int(t)]], device=DEV), mems)
        assert relerr(logits, z[f"step{i}_logits"]) < 2e-2
    assert relerr(mems, z["final_mems"]) < 2e-2
    model.reset_length(4, 6)                                   # same_length with a short memory
    mems = None
    for i in range(3):
        logits, mems = model.forward_generate(torch.from_numpy(z[f"sl_data{i}"]).to(DEV), mems)
        assert relerr(logits, z[f"sl_logits{i}"]) < 2e-2
        assert relerr(mems, z[f"sl_mems{i}"]) < 2e-2


def test_g8_optimizer_steps_vs_reference(golden_dir):
    """clip + Adam + LambdaLR over 4 optimiser steps with batch_chunk 2 (train.py:133-169)."""
    from commu_amd.functional import masked_mean
    from commu_amd.optim import FusedAdam, clip_grad_norm_, lr_lambda_factory
    z = load(golden_dir, "g8_optim.npz")
    model, _ = build_from_fixture(z)
    chunk = int(z["meta"][8])
    lr = 0.004
    opt = FusedAdam(model, lr=lr)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda=lr_lambda_factory(int(z["warmup"]), lr, 0.0001))
    mems = [None] * chunk
    for step in range(int(z["nsteps"])):
        data = torch.from_numpy(z[f"data{step}"]).to(DEV)
        target = torch.from_numpy(z[f"target{step}"]).to(DEV)
        reset = torch.from_numpy(z[f"reset{step}"]).to(DEV)
        assert abs(opt.param_groups[0]["lr"] - float(z[f"lr{step}"])) < 1e-12
        model.zero_grad()
        tot = 0.0
        for i in range(chunk):
            d, t, r = [torch.chunk(x, chunk, dim)[i].contiguous() for x, dim in ((data, 1), (target, 1), (reset, 0))]
            loss, mems[i] = model(d, t, r, mems[i])
            loss = masked_mean(loss, t, 0, 1.0 / chunk)
            loss.backward()
            tot += float(loss)
        gn = clip_grad_norm_(model, float(z["clip"]), opt)
        opt.step()
        opt.zero_grad()
        sched.step()
        assert abs(tot - float(z[f"loss{step}"])) < 1e-2
        assert abs(float(gn) - float(z[f"gnorm{step}"])) < 5e-2 * float(z[f"gnorm{step}"])
    # Adam normalises each coordinate's step to ~lr, so after 4 steps parameters moved by <= ~3*lr_eff;
    # compare the UPDATE (after - before), which is what the optimiser computed
    # (the Adam kernel itself is checked to 1e-6 in test_kernels_gpu.py; here bf16 gradient noise is
    #  amplified by Adam's per-coordinate normalisation wherever |grad| is at the noise level, so the
    #  end-to-end check is on the direction of each tensor's update and of the whole update)
    cos, allu, allr = {}, [], []
    for name, p in model.named_parameters():
        before, after = z["p::" + name], z["after::" + name]
        upd_ref = torch.from_numpy(after - before).flatten()
        upd = (p.detach().cpu() - torch.from_numpy(before)).flatten()
        cos[name] = float(torch.dot(upd, upd_ref) / (upd.norm() * upd_ref.norm() + 1e-30))
        allu.append(upd)
        allr.append(upd_ref)
    allu, allr = torch.cat(allu), torch.cat(allr)
    total = float(torch.dot(allu, allr) / (allu.norm() * allr.norm()))
    _dump("g8", {"cos": cos, "total": total, "norm_ratio": float(allu.norm() / allr.norm())})
    assert total > 0.99, total
    assert abs(float(allu.norm() / allr.norm()) - 1.0) < 0.03
    # per tensor: >= 0.99 everywhere except r_net.weight (>= 0.8).  The bound is derived, not tuned: Adam's first update of a
    # coordinate is -lr * sign(g), so the cosine between two first updates of a tensor IS the mean sign agreement of the two
    # gradients.  test_g1_forward_backward_vs_reference measures that for every tensor (`sign_cos` in its dump): 0.70-0.80 for
    # r_net.weight, >= 0.97 elsewhere -- while the coordinates whose sign agrees carry 99.95 % of that gradient's L1 mass
    # (`sign_mass`, asserted there) and the gradient's own cosine is > 0.995: most of r_net.weight's coordinates (high-frequency
    # sinusoid inputs, summed over a band whose terms cancel) lie below the bf16 rounding noise of dS, and Adam turns each of
    # them into a full-size step of random sign.  After 4 steps the second moments damp that a little: measured 0.85 / 0.91.
    # (Feeding the r_net weight-gradient GEMM an fp32-accurate hi + lo dRd instead of the rounded copy changed neither: the
    #  noise enters with dS, upstream of that sum.)
    for k, v in cos.items():
        assert v > (0.8 if k.endswith("r_net.weight") else 0.99), (k, v)


def test_state_dict_roundtrip_and_no_cpu_path(golden_dir):
    from commu_amd._lib import CommuHipError
    z = load(golden_dir, "g1_train_mem.npz")
    model, _ = build_from_fixture(z)
    sd = model.state_dict()
    for k in z.files:
        if k.startswith("p::"):
            assert k[3:] in sd and tuple(sd[k[3:]].shape) == z[k].shape
    assert sd["crit.out_layers.0.weight"].data_ptr() == sd["word_emb.emb_layers.0.weight"].data_ptr()
    cpu_model = type(model)(model.cfg, [0] * 729)
    with pytest.raises(CommuHipError):
        cpu_model(torch.zeros(4, 2, dtype=torch.long), torch.zeros(4, 2, dtype=torch.long), None, None)


def test_grad_ready_hook_reports_layer_slices_top_down(golden_dir):
    """The overlapped gradient exchange (commu_amd/ddp.py) relies on this contract: after layer i's backward
    has been enqueued, flat_g[lo:hi] covers exactly the parameters of layer i, and layers come top-down."""
    z = load(golden_dir, "g1_train_nomem.npz")
    model, cfg = build_from_fixture(z)
    model.eval()
    data, target = torch.from_numpy(z["data0"]).to(DEV), torch.from_numpy(z["target0"]).to(DEV)
    seen = []
    model.grad_ready_hook = lambda G, lo, hi: seen.append((G.data_ptr(), lo, hi))
    loss, _ = model(data, target, torch.zeros(data.shape[1], dtype=torch.bool, device=DEV), None)
    loss[target != 0].float().mean().backward()
    model.grad_ready_hook = None
    fl = model._ensure_flat()
    L = model.n_layer
    assert len(seen) == L and all(ptr == fl["g"].data_ptr() for ptr, _, _ in seen)
    offs = {n: o for (n, _), o in zip(model.named_parameters(), fl["offs"])}
    for k, (_, lo, hi) in enumerate(seen):
        i = L - 1 - k
        names = [n for n in offs if n.startswith(f"layers.{i}.")]
        assert lo == min(offs[n] for n in names)
        assert hi >= max(offs[n] + dict(model.named_parameters())[n].numel() for n in names)
        nxt = [o for n, o in offs.items() if o >= hi]
        assert not nxt or min(nxt) == hi                     # ends exactly where the next parameter starts
        for n in names:                                      # and the slice aliases the parameters' .grad
            p = dict(model.named_parameters())[n]
            assert p.grad.data_ptr() == fl["g"].data_ptr() + 4 * offs[n]


def test_checkpoint_roundtrip_resumes_identically(tmp_path):
    """save_checkpoint / load_checkpoint (train.py:29-54 layout): model + Adam moments + step + LR schedule.
    A resumed trainer must produce the same parameters as the one that never stopped, and the optimizer state
    must be loadable by torch.optim.Adam (what the reference would do with it)."""
    from commu_amd.model.config_helper import get_cfg
    from commu_amd.model.dataset import BaseVocab, synthetic_batch
    from commu_amd.train import Trainer, build_model, load_checkpoint, save_checkpoint
    cfg = get_cfg(num_layers=2, num_heads=2, units=64, inner_size=128, tgt_length=32, mem_length=32,
                  batch_size=4, batch_chunk=1, dropout=0.0, attention_dropout=0.0)
    dev = torch.device(DEV)
    batches = [synthetic_batch(32, 4, dev, seed=50 + i) for i in range(5)]

    def run(tr, lo, hi):
        for i in range(lo, hi):
            tr.step(*batches[i])

    a = Trainer(build_model(cfg, BaseVocab(), dev, seed=1), cfg)
    run(a, 0, 3)
    path = str(tmp_path / "checkpoint_last.pt")
    save_checkpoint(path, a.model, a.optimizer, BaseVocab(), a.train_step, 1.25, a.scheduler)
    run(a, 3, 5)

    b = Trainer(build_model(cfg, BaseVocab(), dev, seed=99), cfg)          # different init: must be overwritten
    step, best = load_checkpoint(path, b.model, b.optimizer, b.scheduler)
    assert step == 3 and best == 1.25 and b.optimizer.step_count == 3
    b.train_step = step
    b.mems = [None]
    a2 = Trainer(build_model(cfg, BaseVocab(), dev, seed=1), cfg)          # reference run without XL memory carry-over
    run(a2, 0, 3)
    a2.mems = [None]
    run(a2, 3, 5)
    run(b, 3, 5)
    for (n, p), (_, q) in zip(a2.model.named_parameters(), b.model.named_parameters()):
        # (bias gradients are reduced with float atomics: allow last-bit differences)
        assert torch.allclose(p, q, rtol=0, atol=1e-5), n

    ck = torch.load(path, map_location="cpu", weights_only=False)
    ref_params = [torch.nn.Parameter(v.clone()) for k, v in ck["model"].items()
                  if k in dict(a.model.named_parameters())]
    order = [n for n, _ in a.model.named_parameters()]
    ref_params = [torch.nn.Parameter(ck["model"][n].clone()) for n in order]
    adam = torch.optim.Adam(ref_params, lr=0.004)
    adam.load_state_dict(ck["optimizer"])
    st = adam.state[ref_params[3]]
    assert float(st["step"]) == 3.0 and st["exp_avg"].shape == ref_params[3].shape
    assert float(st["exp_avg_sq"].abs().sum()) > 0


def test_all_pad_targets_and_input_check(golden_dir):
    """Edge cases of the loss reduction: a micro-batch whose targets are all pads gives NaN like the reference's
    `loss[target != pad].mean()` of an empty selection, never a crash; `check_inputs` reports ids outside the
    vocabulary as the reference's embedding would (IndexError)."""
    from commu_amd.functional import masked_mean
    z = load(golden_dir, "g1_train_nomem.npz")
    model, cfg = build_from_fixture(z)
    model.eval()
    data = torch.from_numpy(z["data0"]).to(DEV)
    target = torch.zeros_like(data)
    loss, _ = model(data, target, torch.zeros(data.shape[1], dtype=torch.bool, device=DEV), None)
    assert torch.isfinite(loss).all()                       # per-token NLL is still defined
    m = masked_mean(loss, target, 0, 1.0)
    assert torch.isnan(m) and torch.isnan(loss[target != 0].float().mean())
    model.check_inputs = True
    bad = data.clone()
    bad[0, 0] = 729
    with pytest.raises(IndexError):
        model(bad, target, torch.zeros(data.shape[1], dtype=torch.bool, device=DEV), None)


def _g10_cfg(z, same_length):
    from commu_amd.model.config_helper import get_cfg
    L, H, D, DI, T, B, mem_len = [int(x) for x in z["meta"]]
    return get_cfg(num_layers=L, num_heads=H, units=D, inner_size=DI, tgt_length=T, mem_length=mem_len, batch_size=B,
                   batch_chunk=1, dropout=0.0, attention_dropout=0.0, same_length=same_length, warmup_step=2,
                   lr=0.004, clip=1.0)


def test_model_initializer_on_a_reference_written_checkpoint(golden_dir):
    """model_initializer.py:36-51 on a file the reference's train.py wrote: strict=False load, same_length,
    eval, reset_length(1, 4146); generation logits against what the reference computes from the same file."""
    import types
    from commu_amd.midi_generator.model_initializer import ModelInitializeTask
    z = load(golden_dir, "g10_checkpoint.npz")
    task = ModelInitializeTask(types.SimpleNamespace(checkpoint_dir=os.path.join(golden_dir, "g10_checkpoint.pt")),
                               map_location="cpu", device=torch.device(DEV), training_cfg=_g10_cfg(z, False))
    model = task.execute()
    assert model.same_length and not model.training and (model.tgt_len, model.mem_len) == (1, 4146)
    with torch.no_grad():
        logits, mems = model.forward_generate(torch.from_numpy(z["gen_ctx"]).to(DEV), None)
        assert relerr(logits, z["gen_logits0"]) < 3e-2
        for i, t in enumerate(z["gen_tokens"].tolist()):
            logits, mems = model.forward_generate(torch.tensor([[t]], device=DEV), mems)
            assert relerr(logits, z[f"gen_logits{i + 1}"]) < 3e-2, i
        assert mems.shape[1] == z["gen_ctx"].shape[0] + 3


def test_training_resumes_from_the_reference_optimizer_state(golden_dir):
    """Adam moments / step counts and the LambdaLR position written by the reference (torch.optim.Adam's state_dict)
    are taken over by FusedAdam: one more optimiser step moves every tensor the way the reference's next step does."""
    from commu_amd.model.dataset import BaseVocab
    from commu_amd.model.model import MemTransformerLM
    from commu_amd.train import Trainer, read_checkpoint
    z = load(golden_dir, "g10_checkpoint.npz")
    ck = read_checkpoint(os.path.join(golden_dir, "g10_checkpoint.pt"))
    cfg = _g10_cfg(z, False)
    model = MemTransformerLM(cfg, BaseVocab())
    model.load_state_dict(ck["model"])
    model = model.to(DEV).train()
    trainer = Trainer(model, cfg)
    trainer.optimizer.load_state_dict(ck["optimizer"])
    trainer.scheduler.load_state_dict(ck["scheduler"])
    assert abs(trainer.optimizer.param_groups[0]["lr"] - ck["scheduler"]["_last_lr"][0]) < 1e-12
    data, target = torch.from_numpy(z["resume_data"]).to(DEV), torch.from_numpy(z["resume_target"]).to(DEV)
    trainer.step(data, target, torch.zeros(data.shape[1], dtype=torch.bool, device=DEV), int((z["resume_target"] != 0).sum()))
    assert abs(trainer.optimizer.param_groups[0]["lr"] - float(z["resume_lr"])) < 1e-9
    ups, refs = [], []
    for name, p in model.named_parameters():
        before = ck["model"][name].float()
        ups.append((p.detach().cpu() - before).flatten())
        refs.append((torch.from_numpy(z["after::" + name]) - before).flatten())
    u, r = torch.cat(ups), torch.cat(refs)
    cos = float(torch.dot(u, r) / (u.norm() * r.norm()))
    assert cos > 0.97 and abs(float(u.norm() / r.norm()) - 1) < 0.05, (cos, float(u.norm() / r.norm()))


def _write_output_npy(dirname, seqs):
    """The reference's on-disk corpus (preprocessor.py:161-162 / dataset.py:74-87): object arrays of 11 meta ints
    and int16 event arrays."""
    for split, tag in (("train", "train"), ("valid", "val")):
        metas = np.array([np.array(s[:11], dtype=object) for s in seqs[split]], dtype=object)
        ev = np.empty(len(seqs[split]), dtype=object)
        for i, s in enumerate(seqs[split]):
            ev[i] = np.asarray(s[11:]).astype(np.int16)
        np.save(os.path.join(dirname, f"input_{tag}.npy"), metas, allow_pickle=True)
        np.save(os.path.join(dirname, f"target_{tag}.npy"), ev, allow_pickle=True)


def _load_script(name):
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("commu_cli_" + name, os.path.join(root, "commu-code_amd", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_train_cli_runs_on_an_output_npy_directory(tmp_path):
    """BASELINE.json configs[0] plumbing: the reference's `train.py --data_dir --work_dir` flags on a synthetic corpus
    in the on-disk format: steps, logging windows, evaluation, checkpoint_last / checkpoint_best (train.py:113-288)."""
    from commu_amd.train import read_checkpoint
    rng = np.random.RandomState(0)
    corpus = {s: [np.concatenate([rng.randint(560, 729, 11), rng.randint(2, 560, rng.randint(20, 90)), [1]])
                  for _ in range(n)] for s, n in (("train", 40), ("valid", 12))}
    data_dir, work = tmp_path / "output_npy", tmp_path / "work"
    data_dir.mkdir()
    _write_output_npy(str(data_dir), corpus)
    cli = _load_script("train")
    run_dir = cli.main(["--data_dir", str(data_dir), "--work_dir", str(work), "--num_layers", "2", "--num_heads", "2",
                        "--units", "64", "--inner_size", "128", "--tgt_length", "16", "--mem_length", "16",
                        "--batch_size", "4", "--batch_chunk", "2", "--max_step", "6", "--log_interval", "2",
                        "--eval_interval", "3"])
    log = open(os.path.join(run_dir, "train_rank0.log")).read()
    assert log.count("Train Step") == 3 and log.count("Eval step") == 2 and "End of training" in log
    assert os.path.exists(os.path.join(run_dir, "config.yml"))
    for name in ("checkpoint_last.pt", "checkpoint_best.pt"):
        ck = read_checkpoint(os.path.join(run_dir, name))
        assert ck["train_step"] in (3, 6) and ck["amp"] is None and "optimizer" in ck
    nll = float(log.split("nll=")[1].split(",")[0])
    assert 5.0 < nll < 7.5                                  # ~ln(729) = 6.59 at initialisation
    # train.py:486-513: the best checkpoint reloaded into a fresh same_length model and evaluated on the test split
    # (= the validation file, dataset.py:81-86), reported in the reference's closing line
    assert log.count("| End of training | test nll") == 1
    tail = log.split("| End of training | test nll")[1]
    end_nll = float(tail.split("|")[0])
    best = read_checkpoint(os.path.join(run_dir, "checkpoint_best.pt"))
    # the same number as the "Test step" line written when that checkpoint was saved (same weights, same split)
    tests = [float(x.split(",")[0]) for x in log.split("test nll=")[1:]]
    assert abs(end_nll - tests[-1]) <= 6e-3 and abs(end_nll - float(best["best_val_loss"])) <= 6e-3, (end_nll, tests, best["best_val_loss"])


def test_generate_cli_from_a_reference_checkpoint(golden_dir, tmp_path):
    """generate.py's flags end to end on the reference-written checkpoint: model initialiser, meta encoding, parallel
    forced decoding, validators, sequences.json."""
    import json
    z = load(golden_dir, "g10_checkpoint.npz")
    cli = _load_script("generate")
    parsers = cli.parse_args()
    prog = "-".join(["Am"] * 8 + ["G"] * 8 + ["F"] * 8 + ["E"] * 8)
    argv = ["--checkpoint_dir", os.path.join(golden_dir, "g10_checkpoint.pt"), "--output_dir", str(tmp_path / "out"),
            "--bpm", "70", "--audio_key", "aminor", "--time_signature", "4/4", "--pitch_range", "mid_high",
            "--num_measures", "8", "--inst", "acoustic_piano", "--genre", "newage", "--min_velocity", "60",
            "--max_velocity", "80", "--track_role", "main_melody", "--rhythm", "standard", "--chord_progression",
            prog + "-" + prog, "--num_generate", "2", "--max_rounds", "1"]
    margs, _ = parsers["model_args"].parse_known_args(argv)
    iargs, _ = parsers["input_args"].parse_known_args(argv)
    import types
    import commu_amd.midi_generator.model_initializer as mi
    # the fixture checkpoint is a 2-layer model: the initialiser is told its shape (the reference always assumes
    # the default one); generation length shortened through the inference cfg
    orig = mi.get_default_cfg_inference

    def short_cfg():
        c = orig()
        c.defrost()
        c.GENERATION.generation_length = 120
        return c
    mi.get_default_cfg_inference = short_cfg
    try:
        seqs = cli.main(margs, iargs, training_cfg=_g10_cfg(z, False))
    finally:
        mi.get_default_cfg_inference = orig
    out = json.load(open(tmp_path / "out" / "sequences.json"))
    assert out["encoded_meta"] == [574, 623, 627, 635, 639, 642, 651, 684, 694, 720, 727]
    assert out["sequences"] == seqs and len(seqs) <= 2          # a random model rarely passes both validators
    # replicas (num_generate split across GPUs, no collective): two replica PROCESSES, both on the
    # test box's one GPU, shares 2 + 1
    assert cli.split_num_generate(3, 2) == [2, 1] and cli.split_num_generate(2, 8) == [1, 1]
    iargs.num_generate = 3
    seqs2 = cli.main(margs, iargs, training_cfg=_g10_cfg(z, False), device_indices=[0, 0])
    out2 = json.load(open(tmp_path / "out" / "sequences.json"))
    assert out2["sequences"] == seqs2 and len(seqs2) <= 3
    assert out2["encoded_meta"] == out["encoded_meta"]


@pytest.mark.gpu
def test_relu_sign_bits_option_gives_identical_gradients(monkeypatch):
    """model.relu_sign_bits (the default; False: the bf16 activations as the mask): the FF ReLU backward reads one bit per element
    written by the forward GEMM instead of the bf16 activations -- same loss, same gradients (dropout on, same seeds)."""
    from commu_amd.model.config_helper import get_cfg
    from commu_amd.model.dataset import BaseVocab, synthetic_batch
    from commu_amd.train import build_model
    monkeypatch.setenv("COMMU_GEMM8_ALWAYS", "1")          # small test shape: 1024 x 512 outputs are 8 tiles
    dev = torch.device("cuda", 0)
    cfg = get_cfg(num_layers=2, num_heads=4, units=256, inner_size=512, tgt_length=64, mem_length=0, batch_size=16,
                  batch_chunk=1, dropout=0.1, attention_dropout=0.1)
    d, t, r, _ = synthetic_batch(64, 16, dev, seed=3)
    out = []
    for bits in (False, True):
        model = build_model(cfg, BaseVocab(), dev, seed=5)
        model.train()
        model.relu_sign_bits = bits
        torch.manual_seed(11)
        loss, _ = model(d, t, r, None)
        loss.float().mean().backward()
        out.append((loss.detach().clone(), [p.grad.detach().clone() for p in model.parameters()]))
    assert torch.equal(out[0][0], out[1][0])
    for ga, gb in zip(out[0][1], out[1][1]):          # (bias / LayerNorm-parameter column sums use fp32 atomics: order noise)
        assert float((ga - gb).abs().max()) <= 1e-5 * float(ga.abs().max()) + 1e-12
    assert any(float(g.abs().max()) > 0 for g in out[0][1])


@pytest.mark.gpu
def test_residual_gradient_in_layernorm_backward_option():
    """model.resid_in_ln_bwd (opt-in, commu_layernorm_bwd_add): the residual branch's gradient enters the LayerNorm backward as a
    second addend instead of the dX GEMM's epilogue -- the same gradients up to the bf16 rounding of one intermediate (the sum
    is no longer rounded before the LayerNorm backward reads it).  Two segments with XL memory, reset columns, dropout on."""
    from commu_amd.model.config_helper import get_cfg
    from commu_amd.model.dataset import BaseVocab, synthetic_batch
    from commu_amd.train import build_model
    dev = torch.device("cuda", 0)
    cfg = get_cfg(num_layers=3, num_heads=4, units=256, inner_size=512, tgt_length=128, mem_length=64, batch_size=8,
                  batch_chunk=1, dropout=0.1, attention_dropout=0.1)
    segs = [synthetic_batch(128, 8, dev, seed=30 + i, reset_prob=0.3) for i in range(2)]
    out = []
    for flag in (False, True):
        model = build_model(cfg, BaseVocab(), dev, seed=5)
        model.train()
        model.resid_in_ln_bwd = flag
        model.fixed_drop_seed = 77
        mems = None
        model.zero_grad()
        for d, t, r, _ in segs:
            loss, mems = model(d, t, r, mems)
            loss.float().mean().backward()
        out.append([p.grad.detach().clone() for p in model.parameters()])
    for ga, gb in zip(*out):
        assert float((ga - gb).abs().max()) <= 2e-2 * float(ga.abs().max()) + 1e-12          # (measured: 8e-3)
        assert float(torch.dot(ga.flatten(), gb.flatten()) / (ga.norm() * gb.norm() + 1e-30)) > 0.9995


@pytest.mark.gpu
def test_forward_saved_probabilities_option_gives_the_same_step(monkeypatch):
    """ops.FWD_SAVES_P (opt-in): the training forward of d_head 64 saves its probabilities and the query-stationary
    backward kernel reads them instead of recomputing scores.  Two segments with XL memory, reset columns and dropout on
    (same seeds): the loss is identical (same forward arithmetic) and every gradient agrees with the default path to the
    bf16 rounding of one more intermediate (the saved probability)."""
    from commu_amd import ops
    from commu_amd.model.config_helper import get_cfg
    from commu_amd.model.dataset import BaseVocab, synthetic_batch
    from commu_amd.train import build_model
    dev = torch.device("cuda", 0)
    cfg = get_cfg(num_layers=2, num_heads=2, units=128, inner_size=256, tgt_length=96, mem_length=64, batch_size=6,
                  batch_chunk=1, dropout=0.1, attention_dropout=0.1)
    segs = [synthetic_batch(96, 6, dev, seed=30 + i, reset_prob=0.3) for i in range(2)]
    out = []
    for flag in (False, True):
        monkeypatch.setattr(ops, "FWD_SAVES_P", flag)
        model = build_model(cfg, BaseVocab(), dev, seed=5)
        model.train()
        model.fixed_drop_seed = 4242
        mems, losses = None, []
        model.zero_grad()
        for d, t, r, _ in segs:
            loss, mems = model(d, t, r, mems)
            loss.float().mean().backward()
            losses.append(loss.detach().clone())
        out.append((losses, {n: p.grad.detach().clone() for n, p in model.named_parameters()}))
    for la, lb in zip(out[0][0], out[1][0]):
        assert torch.equal(la, lb)
    for n in out[0][1]:
        ga, gb = out[0][1][n].float(), out[1][1][n].float()
        cos = float((ga * gb).sum() / (ga.norm() * gb.norm() + 1e-30))
        assert cos > 0.9995, (n, cos)
        assert float((ga - gb).abs().max()) <= 2e-2 * float(ga.abs().max()) + 1e-12, n


@pytest.mark.gpu
def test_persistent_attention_scratch_under_changing_reset_patterns():
    """The dS-by-distance / P scratch lives across layers and steps and is never cleared; with reset_mems the kernel
    zero-writes the distances of the memory tiles a fresh sequence skips.  Ten steps with a different reset pattern each
    (XL memory carried over) must give the gradients of a run whose scratch is fresh and NaN-poisoned at every call."""
    from commu_amd import ops
    from commu_amd.model.config_helper import get_cfg
    from commu_amd.model.dataset import BaseVocab, synthetic_batch
    from commu_amd.train import build_model
    dev = torch.device("cuda", 0)
    T, M, B = 64, 128, 8
    cfg = get_cfg(num_layers=2, num_heads=2, units=128, inner_size=256, tgt_length=T, mem_length=M, batch_size=B,
                  batch_chunk=1, dropout=0.0, attention_dropout=0.0)
    gen = torch.Generator().manual_seed(17)
    steps = []
    for i in range(10):
        d, t, _, _ = synthetic_batch(T, B, dev, seed=100 + i)
        r = (torch.rand(B, generator=gen) < 0.4).to(dev)
        if i == 3:
            r[:] = False
        if i == 6:
            r[:] = True
        steps.append((d, t, r))
    grads = []
    for poison in (False, True):
        model = build_model(cfg, BaseVocab(), dev, seed=9)
        model.train()
        mems, run = None, []
        ops.POISON_SCRATCH = poison
        try:
            for d, t, r in steps:
                model.zero_grad()
                loss, mems = model(d, t, r, mems)
                loss.float().mean().backward()
                run.append([p.grad.detach().clone() for p in model.parameters()])
        finally:
            ops.POISON_SCRATCH = False
        grads.append(run)
    for sa, sb in zip(*grads):
        for ga, gb in zip(sa, sb):
            assert torch.isfinite(ga).all() and torch.isfinite(gb).all()
            assert float((ga - gb).abs().max()) <= 1e-5 * float(ga.abs().max()) + 1e-10


@pytest.mark.parametrize("tag", ["mem", "nomem", "dh50"])
def test_g1_gradients_with_the_builds_relu_gates_injected(golden_dir, tag):
    """Tight window for the FFN gradients (test_g1_forward_backward_vs_reference allows the first FFN Linear 0.35: a
    pre-activation within bf16 rounding of 0 flips its ReLU gate and with it a whole term of that unit's gradient).
    Here the oracle runs the same three segments with the BUILD's gates injected (as the dropout tests inject the
    build's masks): what is left is rounding only, and pos_ff.CoreNet.0 is held to the 6e-2 of every other tensor."""
    from oracle import xl_ref as X
    z = load(golden_dir, f"g1_train_{tag}.npz")
    model, cfg = build_from_fixture(z)
    model.eval()
    model.keep_saved = True
    params = {k[3:]: torch.from_numpy(z[k]).clone().requires_grad_(True) for k in z.files
              if k.startswith("p::") and k[3:] != "crit.out_layers.0.weight" and "inv_freq" not in k}
    L, H, D, DI = (int(cfg.MODEL.num_layers), int(cfg.MODEL.num_heads), int(cfg.MODEL.units), int(cfg.MODEL.inner_size))
    s = X.XLShape(L, H, D, DI)
    mem_len = int(cfg.TRAIN.mem_length)
    mems, omems = None, None
    model.zero_grad()
    flips = 0
    for seg in range(3):
        data, target, reset = (torch.from_numpy(z[f"{k}{seg}"]) for k in ("data", "target", "reset"))
        loss, mems = model(data.to(DEV), target.to(DEV), reset.to(DEV), mems)
        sv = model.last_saved
        T, B = data.shape
        gates = [(sv.hid[i].float().view(T, B, -1)[..., :DI] > 0).float().cpu() for i in range(L)]

        def drop(site, x):
            return x

        def relu_gate(li, zz):
            nonlocal flips
            flips += int(((zz > 0).float() != gates[li]).sum())
            return gates[li]
        drop.relu_gate = relu_gate
        nll, omems = X.forward_loss(params, s, data, target, reset, omems, mem_len, False, drop)
        if omems is not None:
            omems = omems.detach()
        nll[target != 0].float().mean().backward()
        loss[target.to(DEV) != 0].float().mean().backward()
    worst = {}
    for name, p in model.named_parameters():
        if name not in params:
            continue
        worst[name] = relerr(p.grad, params[name].grad)
    _dump(f"g1_gates_{tag}", {"relerr": worst, "flipped_gates": flips})
    assert flips > 0                                   # (the fixture does contain near-zero pre-activations)
    for k, v in worst.items():
        assert v < 6e-2, (k, v, flips)
