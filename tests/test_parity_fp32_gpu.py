"""fp32 PARITY MODE of the generation path (model.parity_fp32 / generate.py --parity; csrc/parity_f32.hip).

The reference computes in fp32 throughout (train.py:48 `amp = None`); BASELINE.json's north star asks for bit-exact greedy
tokens.  The bf16 throughput path can only promise that where the top-1 / top-2 gap exceeds its logit error
(tests/test_decode_gpu.py); this mode only needs the gap to exceed fp32 summation-order noise (~1e-6 of the range): fp32 operands end to end, checked here
against the reference's own fixtures (G2 logits / memories, G6 greedy traces) and against the oracle on random-init models
with NO engineered output bias, free-running for 256 greedy steps.
Tolerances: kernels <= 2e-6 of range against float64; logits <= 1e-4 of range (measured ~1e-6) against the fp32 oracle."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import xl_ref as X  # noqa: E402

DEV = "cuda"


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def rel(a, b):
    a = torch.as_tensor(a).double().cpu()
    b = torch.as_tensor(b).double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.mark.parametrize("M,N,K", [(65, 129, 68), (1, 729, 512), (64, 1536, 512), (1000, 1024, 500), (130, 500, 1000)])
def test_linear_f32_vs_float64(M, N, K):
    from commu_amd import ops
    g = torch.Generator().manual_seed(M * 7 + N)
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) * 0.05
    bias = torch.randn(N, generator=g)
    R = torch.randn(M, N, generator=g)
    want = A.double() @ W.double().t()
    got = ops.gemm_nt_f32(A.to(DEV), W.to(DEV))
    assert rel(got, want) < 2e-6
    want2 = torch.relu(want + bias.double()) + R.double()
    got2 = ops.gemm_nt_f32(A.to(DEV), W.to(DEV), bias=bias.to(DEV), resid=R.to(DEV), relu=True)
    assert rel(got2, want2) < 2e-6
    # strided views (the k | v thirds of a projection weight, a column slice of the output): the unaligned path
    big = torch.zeros(M, N + 3, device=DEV)
    ops.gemm_nt_f32(A.to(DEV)[:, 1:], W.to(DEV)[:, 1:], out=big[:, 3:])
    assert rel(big[:, 3:], A[:, 1:].double() @ W[:, 1:].double().t()) < 2e-6 and float(big[:, :3].abs().max()) == 0.0


def test_embedding_sinusoid_layernorm_f32():
    from commu_amd import ops
    g = torch.Generator().manual_seed(3)
    E = torch.randn(729, 500, generator=g)
    tok = torch.randint(0, 729, (37,), generator=g)
    assert torch.equal(ops.embed_f32(tok.to(DEV), E.to(DEV)).cpu(), E[tok] * (500 ** 0.5))
    # model.py:142-147 with positions = distances 0 .. n-1: torch's own fp32 sin / cos of the fp32 outer product
    D, n = 512, 4147
    inv_freq = 1 / (10000 ** (torch.arange(0.0, D, 2.0) / D))
    sinus = torch.ger(torch.arange(0.0, n), inv_freq)
    want = torch.cat([sinus.sin(), sinus.cos()], dim=-1)
    got = ops.posemb_f32(inv_freq.to(DEV), n, D).cpu()
    assert float((got - want).abs().max()) < 4e-7          # (a few ulp of values <= 1: two correct sine implementations)
    x = torch.randn(70, 500, generator=g) * 3 + 1
    gam, bet = torch.randn(500, generator=g), torch.randn(500, generator=g)
    want = torch.nn.functional.layer_norm(x.double(), (500,), gam.double(), bet.double(), 1e-5)
    assert rel(ops.layernorm_f32(x.to(DEV), gam.to(DEV), bet.to(DEV), 1e-5), want) < 2e-6


def _model(L, H, D, DI, seed, std=0.02, mem_len=4146):
    from test_configs_gpu import build
    model, cfg, s, params = build(L, H, D, DI, 1, mem_len, seed=seed)
    if std != 0.02:
        g = torch.Generator().manual_seed(seed + 100)
        with torch.no_grad():
            for k in list(params):
                if "layer_norm" in k or k.endswith(".bias") or "inv_freq" in k:
                    continue
                params[k] = torch.randn(params[k].shape, generator=g) * std
            sd = {k: v.clone() for k, v in params.items()}
            sd["crit.out_layers.0.weight"] = sd["word_emb.emb_layers.0.weight"]
            model.load_state_dict(sd, strict=False)
    model.eval()
    model.same_length = True
    model.reset_length(1, mem_len)
    model.parity_fp32 = True
    return model, s, params


@pytest.mark.parametrize("shape", [(6, 8, 512, 1024), (6, 10, 500, 1000)], ids=["L6_D512_dh64", "L6_D500_dh50"])
def test_forward_generate_f32_vs_oracle(shape):
    """model.forward_generate in parity mode against oracle.xl_ref.forward_generate (fp32 PyTorch on the CPU): a context, a
    multi-token segment over that memory, single-token steps, and the same_length mask with a SHORT memory (the window
    slides: model.py:549-568).  Logits and memories within 1e-4 of their range (the achieved error is printed)."""
    L, H, D, DI = shape
    model, s, params = _model(L, H, D, DI, seed=23)
    g = torch.Generator().manual_seed(2)
    worst = 0.0
    mems, omems = None, None
    for T in (48, 7, 1, 1):
        data = torch.randint(1, 729, (T, 3), generator=g)
        with torch.no_grad():
            ref, omems = X.forward_generate(params, s, data, omems, 4146, True)
        logits, mems = model.forward_generate(data.to(DEV), mems)
        assert logits.dtype == torch.float32 and mems.dtype == torch.float32
        assert tuple(logits.shape) == tuple(ref.shape) and tuple(mems.shape) == tuple(omems.shape)
        worst = max(worst, rel(logits, ref), rel(mems, omems))
    model.reset_length(4, 6)          # every query sees exactly 6 keys once the memory is full
    mems, omems = None, None
    for _ in range(4):
        data = torch.randint(1, 729, (4, 2), generator=g)
        with torch.no_grad():
            ref, omems = X.forward_generate(params, s, data, omems, 6, True)
        logits, mems = model.forward_generate(data.to(DEV), mems)
        worst = max(worst, rel(logits, ref), rel(mems, omems))
    print(f"fp32 parity forward_generate {shape}: worst logit / memory error {worst:.2e} of range")
    assert worst < 1e-4


def test_clamp_len_fixture_in_parity_mode(golden_dir):
    """cfg.MODEL.clamp_len = 9 (model.py:581-582): the reference's OWN three segments (fixture g1_train_clamp) through
    forward_generate in parity mode -- the memories it returns are the reference's to 1e-5 of their range (with the clamp
    ignored they differ by ~1e-3: the bf16 tolerances could not tell), and the logits equal the oracle's with the clamp."""
    from test_model_gpu import build_from_fixture
    from test_oracle_golden import params_of, shape_of
    z = load(golden_dir, "g1_train_clamp.npz")
    model, _ = build_from_fixture(z)
    assert model.clamp_len == 9
    model.eval()
    model.parity_fp32 = True
    s, p = shape_of(z["meta"]), params_of(z)
    mems, omems, omems_noclamp = None, None, None
    for seg in range(2):          # (segment 2 resets a column: forward_generate has no reset_mems)
        data = torch.from_numpy(z[f"data{seg}"])
        with torch.no_grad():
            ref, omems = X.forward_generate(p, s, data, omems, 16, False, clamp_len=9)
            other, omems_noclamp = X.forward_generate(p, s, data, omems_noclamp, 16, False)
        logits, mems = model.forward_generate(data.to(DEV), mems)
        assert rel(logits, ref) < 1e-5 and rel(mems, z[f"mems{seg}"]) < 1e-5
        assert rel(other, ref) > 1e-4          # the fixture does exercise the clamp


def test_g2_forward_generate_fixture_in_parity_mode(golden_dir):
    """The reference's OWN outputs (fixture G2: context, four single-token steps, same_length with a short memory), which the
    bf16 path meets to 2e-2 of range, met to 1e-5 in parity mode."""
    from test_model_gpu import build_from_fixture
    z = load(golden_dir, "g2_generate.npz")
    model, _ = build_from_fixture(z)
    model.eval()
    model.reset_length(1, 4146)
    model.parity_fp32 = True
    logits, mems = model.forward_generate(torch.from_numpy(z["ctx"]).to(DEV), None)
    assert rel(logits, z["ctx_logits"]) < 1e-5 and tuple(mems.shape) == z["ctx_mems"].shape
    assert rel(mems, z["ctx_mems"]) < 1e-5
    for i, t in enumerate(z["toks"]):
        logits, mems = model.forward_generate(torch.tensor([[int(t)]], device=DEV), mems)
        assert rel(logits, z[f"step{i}_logits"]) < 1e-5
    assert rel(mems, z["final_mems"]) < 1e-5
    model.reset_length(4, 6)
    mems = None
    for i in range(3):
        logits, mems = model.forward_generate(torch.from_numpy(z[f"sl_data{i}"]).to(DEV), mems)
        assert rel(logits, z[f"sl_logits{i}"]) < 1e-5 and rel(mems, z[f"sl_mems{i}"]) < 1e-5


@pytest.mark.parametrize("std", [0.02, 0.09], ids=["init_std_0.02", "init_std_0.09"])
def test_256_greedy_steps_token_exact_without_bias_engineering(std):
    """L6 D512, random initialisation, NO engineered output bias: an 11-token context, then 256 FREE-RUNNING greedy steps --
    the K/V-cache decode step in parity mode feeds on its own argmax, the oracle (forward_generate over its growing fp32
    memory, like midi_inferrer.py:199-207) on its own -- must produce identical token sequences for every sequence of
    the batch.  Reported: the smallest top-1 / top-2 logit gap the oracle saw, the largest logit error, distinct tokens.
    std 0.02: the usual small init (the tied embedding makes a token's own logit dominate -- a degenerate but honest
    trajectory); std 0.09: weights large enough for the trajectory to wander over the vocabulary."""
    from commu_amd.generate import DecodeState
    B, T0, NSTEP = 3, 11, 256
    model, s, params = _model(6, 8, 512, 1024, seed=77, std=std)
    g = torch.Generator().manual_seed(1)
    ctx = torch.randint(2, 729, (T0, B), generator=g)
    ctx[0] = 0
    with torch.no_grad():
        ref, omems = X.forward_generate(params, s, ctx, None, 4146, True)
    st = DecodeState(model, B, T0 + NSTEP + 8)
    assert st.parity and st.kc.dtype == torch.float32
    st.prefill(ctx.to(DEV))
    # the first token comes from the context's last logits row: the reference-API path in parity mode
    logits0, _ = model.forward_generate(ctx.to(DEV), None)
    otok = ref[-1].argmax(-1)
    tok = logits0[-1].argmax(-1)
    assert torch.equal(tok.cpu(), otok)
    ones = torch.ones(B, dtype=torch.uint8, device=DEV)
    min_gap, worst, rng = float("inf"), 0.0, float(ref.abs().max())
    seq, oseq = [tok.cpu().tolist()], [otok.tolist()]
    for step in range(NSTEP):
        with torch.no_grad():
            ref, omems = X.forward_generate(params, s, otok[None], omems, 4146, True)
        lg = st.step(tok, ones, ones)[:, :729]
        top2 = ref[0].topk(2, dim=-1).values
        min_gap = min(min_gap, float((top2[:, 0] - top2[:, 1]).min()))
        if seq == oseq:          # (the logits are comparable while the two trajectories agree)
            worst = max(worst, float((lg.cpu() - ref[0]).abs().max()) / rng)
        otok = ref[0].argmax(-1)
        tok = lg.argmax(-1)
        seq.append(tok.cpu().tolist())
        oseq.append(otok.tolist())
    distinct = len({t for row in oseq for t in row})
    print(f"fp32 parity greedy decode (init std {std}): {NSTEP} steps x {B} sequences, {distinct} distinct tokens, min top1-top2 gap "
          f"{min_gap:.3e}, worst logit error {worst:.2e} of range {rng:.2f}")
    assert seq == oseq
    assert worst < 1e-4
    assert int(st.klen[0]) == T0 + NSTEP


@pytest.mark.parametrize("tags", [("greedy8",), ("greedy5",), ("greedy8", "greedy5")])
def test_g6_greedy_reference_traces_in_parity_mode(golden_dir, tags):
    """The reference's greedy traces (fixture G6: tokens, model-step trace with quirks Q3 / Q4, chord forcing) through the
    device-resident decode loop in PARITY mode, eagerly for one sequence and as the captured hipGraph for two in parallel:
    token- and trace-exact, as on the bf16 path (tests/test_decode_gpu.py)."""
    import test_decode_gpu as TD
    from commu_amd.generate import BatchedGenerator
    z = load(golden_dir, "g6_decode.npz")
    model = TD._build(golden_dir, z, z[f"{tags[0]}_bias"])
    model.parity_fp32 = True
    if len(tags) == 1:
        tag = tags[0]
        glen = int(z[f"{tag}_cfg"][3])
        dec = TD._decoder(model, z, [tag])
        assert dec.state.parity
        with torch.no_grad():
            for _ in range(glen + 1):
                dec.iteration()
                if int(dec.fsm[0, 5]):
                    break
        seqs, traces = dec.sequences()
        assert seqs[0] == z[f"{tag}_seq"].tolist()
        assert traces[0] == [tuple(t) for t in z[f"{tag}_trace"].tolist()]
        return
    temp, _, top_k, _ = z[f"{tags[0]}_cfg"]
    glen = max(int(z[f"{t}_cfg"][3]) for t in tags)
    gen = BatchedGenerator(model, torch.device(DEV), generation_length=glen)
    gen.trace = [[] for _ in tags]
    seqs, _ = gen.generate([z["encoded_meta"].tolist()] * len(tags), [TD._data(z, t) for t in tags], float(temp), int(top_k))
    for b, tag in enumerate(tags):
        ref = z[f"{tag}_seq"].tolist()
        ref_trace = [tuple(t) for t in z[f"{tag}_trace"].tolist()]
        assert seqs[b][:len(ref)] == ref, tag
        assert gen.trace[b][:len(ref_trace)] == ref_trace, tag


def test_generate_cli_parity_flag(golden_dir, tmp_path):
    """`generate.py --parity` end to end on the reference-written checkpoint: the model initialiser switches the model to the
    fp32 mode, the decode state it builds is the fp32 one, sequences.json is written.  Greedy (--temperature 0) so that the run
    is deterministic: two invocations give the same sequences."""
    import json
    import test_model_gpu as TM
    z = load(golden_dir, "g10_checkpoint.npz")
    cli = TM._load_script("generate")
    parsers = cli.parse_args()
    prog = "-".join(["Am"] * 8 + ["G"] * 8 + ["F"] * 8 + ["E"] * 8)
    argv = ["--checkpoint_dir", os.path.join(golden_dir, "g10_checkpoint.pt"), "--parity", "--output_dir", str(tmp_path / "out"),
            "--bpm", "70", "--audio_key", "aminor", "--time_signature", "4/4", "--pitch_range", "mid_high",
            "--num_measures", "8", "--inst", "acoustic_piano", "--genre", "newage", "--min_velocity", "60",
            "--max_velocity", "80", "--track_role", "main_melody", "--rhythm", "standard", "--chord_progression",
            prog + "-" + prog, "--num_generate", "1", "--max_rounds", "1", "--temperature", "0", "--gpus", "1"]
    margs, _ = parsers["model_args"].parse_known_args(argv)
    iargs, _ = parsers["input_args"].parse_known_args(argv)
    assert margs.parity is True
    import commu_amd.midi_generator.model_initializer as mi
    from commu_amd import generate as G
    orig = mi.get_default_cfg_inference
    built = []
    orig_init = G.DecodeState.__init__

    def spy(self, model, B, Lmax):
        orig_init(self, model, B, Lmax)
        built.append(self.parity)

    def short_cfg():
        c = orig()
        c.defrost()
        c.GENERATION.generation_length = 64
        return c
    mi.get_default_cfg_inference = short_cfg
    G.DecodeState.__init__ = spy
    try:
        outs = []
        for _ in range(2):
            cli.main(margs, iargs, training_cfg=TM._g10_cfg(z, False))
            outs.append(json.load(open(tmp_path / "out" / "sequences.json")))
    finally:
        mi.get_default_cfg_inference = orig
        G.DecodeState.__init__ = orig_init
    assert built and all(built)
    assert outs[0] == outs[1] and outs[0]["encoded_meta"][0] == 574


# ---------------------------------------------------------------------------------------------- forward (loss) / evaluate
@pytest.mark.parametrize("tag", ["mem", "nomem", "dh50", "clamp"])
def test_g1_loss_and_memory_fixtures_in_parity_mode(golden_dir, tag):
    """model.parity_fp32 also serves forward() under no_grad (model.py:678-693: per-token NLL + new memory): the reference's
    own outputs for three consecutive segments -- memory carried, a reset_mems column, pad tails, d_head 50, clamp_len 9 --
    to fp32 summation-order accuracy (bf16 path: 4e-2 on the worst token)."""
    from test_model_gpu import build_from_fixture
    z = load(golden_dir, f"g1_train_{tag}.npz")
    model, cfg = build_from_fixture(z)
    model.eval()
    model.parity_fp32 = True
    mems = None
    with torch.no_grad():
        for seg in range(3):
            data = torch.from_numpy(z[f"data{seg}"]).to(DEV)
            target = torch.from_numpy(z[f"target{seg}"]).to(DEV)
            reset = torch.from_numpy(z[f"reset{seg}"]).to(DEV)
            loss, mems = model(data, target, reset, mems)
            ref = torch.from_numpy(z[f"loss{seg}"])
            assert loss.dtype == torch.float32 and tuple(loss.shape) == tuple(ref.shape)
            err = float((loss.cpu() - ref).abs().max())
            assert err < 1e-5 * float(ref.abs().max()), (tag, seg, err)          # per-token NLL ~ 6.6: <= 7e-5 absolute
            if tag != "nomem":
                assert mems.dtype == torch.float32 and mems.shape == z[f"mems{seg}"].shape
                assert rel(mems, z[f"mems{seg}"]) < 1e-5, (tag, seg)
            else:
                assert mems is None
            scalar = float(loss[target != 0].mean())
            assert abs(scalar - float(z[f"scalar{seg}"])) < 1e-5 * abs(float(z[f"scalar{seg}"]))
    # forward-only: a gradient-enabled call says so instead of silently leaving the mode
    from commu_amd._lib import CommuHipError
    with pytest.raises(CommuHipError):
        model(data, target, reset, None)


def test_evaluate_in_parity_mode_vs_oracle():
    """Trainer.evaluate (train.py:74-110) with model.parity_fp32: same_length masks, the EVALUATE memory overflowing its window,
    a memory reset between segments -- NLL per token within 1e-6 relative of the fp32 oracle (bf16 path: 6e-3 absolute)."""
    from test_configs_gpu import build
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
        segs.append((data, target, i == 2, int((target != 0).sum())))

    def eval_iter():
        for d, t, r, n in segs:
            yield d.to(DEV), t.to(DEV), r, n
    model.parity_fp32 = True
    tok, nll = Trainer(model, cfg).evaluate(eval_iter)
    ref_tok, ref_nll, omems = 0, 0.0, None
    with torch.no_grad():
        for d, t, r, n in segs:
            if r:
                omems = None
            loss, omems = X.forward_loss(params, s, d, t, None, omems, 80, True)
            ref_nll += n * float(loss[t != 0].double().mean())
            ref_tok += n
    assert tok == ref_tok
    assert abs(nll / tok - ref_nll / ref_tok) < 1e-6 * (ref_nll / ref_tok), (nll / tok, ref_nll / ref_tok)
