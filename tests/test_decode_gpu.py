"""GPU parity of the sampling step and the chord-forced decode loop against fixtures produced by
the reference (g9_sampling, g6_decode) and the oracle's draw."""
import os
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import decode_ref as Dz  # noqa: E402

DEV = "cuda"


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_sampling_probs_match_reference(golden_dir):
    """calc_probs + apply_sampling incl. the compounded temperature of consecutive rejections (Q5),
    the pad column (Q6) and greedy one-hot."""
    from commu_amd import ops
    z = load(golden_dir, "g9_sampling.npz")
    for c in range(int(z["ncase"])):
        row = torch.from_numpy(z[f"c{c}_logits"]).clone()[None].to(DEV)      # [1, 729], modified in place
        temp = float(z[f"c{c}_temp"])
        for r in range(int(z[f"c{c}_rounds"])):
            wrong_ids = z[f"c{c}_r{r}_wrong"].tolist()
            wrong = torch.zeros(1, 729, dtype=torch.uint8)
            wrong[0, wrong_ids] = 1
            probs = torch.empty(1, 729, device=DEV)
            u = torch.tensor([0.5], device=DEV)
            tok = ops.sample_topk(row, temp, 32, wrong=wrong.to(DEV), uniforms=u, probs_out=probs)
            ref = z[f"c{c}_r{r}_probs"]
            got = probs[0].cpu().numpy()
            assert np.abs(got - ref).max() < 2e-6, (c, r)
            assert ((got > 0) == (ref > 0)).all() and got[0] == 0
            assert int(tok) == Dz.draw_inverse_cdf(torch.from_numpy(ref), 0.5)


def test_sampling_topk_selection_matches_torch_topk():
    """The kept set is torch.topk's (radix-select threshold + ties in id order), with and without a rejection mask; a zero
    mask changes nothing; rejected tokens lose their mass and the rest is renormalised."""
    from commu_amd import ops
    g = torch.Generator().manual_seed(7)
    for trial in range(3):
        logits = torch.randn(64, 729, generator=g) * (1 + 2 * trial)
        p = torch.softmax(torch.cat([torch.full((64, 1), float("-inf")), logits[:, 1:] / 0.95], 1), 1)
        top = p.topk(32, dim=1).indices
        ref_keep = torch.zeros(64, 729, dtype=torch.bool).scatter_(1, top, True)
        outs = []
        for mode in ("none", "zeros", "reject"):
            wrong = None
            if mode != "none":
                wrong = torch.zeros(64, 729, dtype=torch.uint8)
                if mode == "reject":
                    wrong[torch.arange(64), top[:, 0]] = 1          # reject every row's most likely token
                    wrong[:, 5] = 1
            pr = torch.zeros(64, 729, device=DEV)
            ops.sample_topk(logits.clone().to(DEV), 0.95, 32, wrong=None if wrong is None else wrong.to(DEV),
                            uniforms=torch.full((64,), 0.5, device=DEV), probs_out=pr)
            outs.append(pr.cpu())
            want = p * ref_keep
            if mode == "reject":
                want = want * (wrong == 0)
            want = want / want.sum(1, keepdim=True)
            assert torch.equal(pr.cpu() > 0, want > 0), (trial, mode)
            assert float((pr.cpu() - want).abs().max()) < 2e-6, (trial, mode)
        assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("top_p", [0.3, 0.6, 0.9, 0.999])
def test_sampling_nucleus_filter_vs_oracle(top_p):
    """The extra top-p mode (commu_sample_topk_topp; the reference has top-k only): the kept set and the renormalised
    probabilities against oracle.decode_ref.apply_top_p applied to the oracle's top-k / rejected-token distribution, with
    and without rejected tokens, including rows with exact ties at the nucleus boundary; the draw is the inverse CDF of
    that distribution.  Rows whose cumulative mass passes within 1e-5 of top_p are checked up to that one token."""
    from commu_amd import ops
    g = torch.Generator().manual_seed(11)
    logits = torch.randn(64, 729, generator=g) * 2.5
    logits[0, 10:40] = 6.0                                   # a plateau of equal probabilities: ties at the boundary
    logits[1, 100:103] = 7.0
    wrong = torch.zeros(64, 729, dtype=torch.uint8)
    wrong[::3, 7] = 1
    wrong[5, logits[5, 1:].argmax() + 1] = 1
    us = torch.rand(64, generator=g)
    for use_wrong in (False, True):
        pr = torch.zeros(64, 729, device=DEV)
        tok = ops.sample_topk(logits.clone().to(DEV), 0.95, 32, wrong=wrong.to(DEV) if use_wrong else None,
                              uniforms=us.to(DEV), probs_out=pr, top_p=top_p).cpu()
        pr = pr.cpu()
        for b in range(64):
            wl = [int(i) for i in wrong[b].nonzero().flatten()] if use_wrong else []
            base = Dz.apply_sampling(Dz.calc_probs(logits[b, 1:].clone(), 0.95), 32, wl)
            ref = Dz.apply_top_p(base, top_p)
            srt = torch.sort(base.double(), descending=True, stable=True).values
            before = torch.cumsum(srt, 0) - srt
            near = bool(((before - top_p).abs() < 1e-5).any())          # the boundary falls on a rounding
            same_set = torch.equal(pr[b] > 0, ref > 0)
            assert same_set or (near and int(((pr[b] > 0) != (ref > 0)).sum()) == 1), (b, use_wrong)
            if same_set:
                assert float((pr[b] - ref).abs().max()) < 2e-6, (b, use_wrong)
                assert abs(float(pr[b].sum()) - 1.0) < 1e-5
                want = Dz.draw_inverse_cdf(ref, float(us[b]))
                if int(tok[b]) != want:
                    cdf = torch.cumsum(ref.double(), 0)
                    assert float((cdf - float(us[b])).abs().min()) < 1e-5
    # top_p = 1 is the reference's step, bit for bit
    a, b_ = torch.zeros(64, 729, device=DEV), torch.zeros(64, 729, device=DEV)
    ops.sample_topk(logits.clone().to(DEV), 0.95, 32, uniforms=us.to(DEV), probs_out=a)
    ops.sample_topk(logits.clone().to(DEV), 0.95, 32, uniforms=us.to(DEV), probs_out=b_, top_p=1.0)
    assert torch.equal(a, b_)


def test_sampling_draw_is_inverse_cdf():
    from commu_amd import ops
    g = torch.Generator().manual_seed(5)
    logits = torch.randn(64, 729, generator=g) * 2
    us = torch.rand(64, generator=g)
    dev_logits = logits.clone().to(DEV)
    probs = torch.empty(64, 729, device=DEV)
    tok = ops.sample_topk(dev_logits, 0.95, 32, uniforms=us.to(DEV), probs_out=probs).cpu()
    probs = probs.cpu()
    for b in range(64):
        ref = Dz.apply_sampling(Dz.calc_probs(logits[b, 1:].clone(), 0.95), 32, [])
        assert float((probs[b] - ref).abs().max()) < 2e-6
        want = Dz.draw_inverse_cdf(ref, float(us[b]))
        if int(tok[b]) != want:                       # only allowed within rounding of a CDF step
            cdf = torch.cumsum(ref.double(), 0)
            assert float((cdf - float(us[b])).abs().min()) < 1e-5
    # active mask: inactive sequences are left untouched
    act = torch.zeros(64, dtype=torch.uint8)
    act[::2] = 1
    before = dev_logits.clone()
    t2 = torch.full((64,), -7, dtype=torch.int32, device=DEV)
    ops.sample_topk(dev_logits, 0.95, 32, uniforms=us.to(DEV), active=act.to(DEV), token=t2)
    assert torch.equal(dev_logits[1::2], before[1::2]) and bool((t2[1::2] == -7).all())
    assert not torch.equal(dev_logits[::2], before[::2])


def test_greedy_on_rejected_token_fails_like_reference():
    """Q12: temperature 0 and the argmax token is rejected -> nothing can be drawn."""
    from commu_amd import ops
    row = torch.zeros(1, 729)
    row[0, 200] = 5.0
    wrong = torch.zeros(1, 729, dtype=torch.uint8)
    wrong[0, 200] = 1
    tok = ops.sample_topk(row.to(DEV), 0.0, 32, wrong=wrong.to(DEV))
    assert int(tok) == -1
    tok = ops.sample_topk(row.to(DEV), 0.0, 32)
    assert int(tok) == 200


def _build(golden_dir, z, bias):
    from test_model_gpu import build_from_fixture
    model, _ = build_from_fixture(z, same_length=True)
    with torch.no_grad():
        model.crit.out_layers[0].bias.copy_(torch.from_numpy(bias).to(DEV))
    model.eval()
    model.reset_length(1, 4146)
    return model


def _data(z, tag):
    return types.SimpleNamespace(num_measures=float(z[f"{tag}_cfg"][1]), chord_token_components={
        "chord_token": z[f"{tag}_chord_token"].tolist(), "chord_position": z[f"{tag}_chord_position"].tolist()})


def _decoder(model, z, tags, record_trace=True, top_p=1.0):
    """A ForcedDecoder loaded with one sequence per fixture tag (same temperature / top-k / bias)."""
    from commu_amd.generate import ForcedDecoder
    temp, _, top_k, _ = z[f"{tags[0]}_cfg"]
    glen = max(int(z[f"{t}_cfg"][3]) for t in tags)
    dec = ForcedDecoder(model, len(tags), glen, 4146, float(temp), int(top_k), record_trace=record_trace, top_p=top_p)
    uni = np.full((len(tags), dec.ld_u), 0.5, dtype=np.float32)
    for b, t in enumerate(tags):
        u = z[f"{t}_uniforms"]
        uni[b, :len(u)] = u
    meta = z["encoded_meta"].tolist()
    dec.load([meta] * len(tags), [_data(z, t) for t in tags], uni)
    return dec


@pytest.mark.parametrize("tag", ["greedy8", "greedy5", "sample8", "sample4x", "sample8m"])
def test_decode_loop_vs_reference_trace(golden_dir, tag):
    """One sequence through the device-resident loop (decide -> K/V-cache step -> sampling kernel -> book-keeping,
    run eagerly so that every draw can be inspected) against the reference's sequential trace.
    greedy and the margin-enforced sampled fixture: token-exact, model-step trace exact (quirks Q3/Q4/Q5).
    other sampled fixtures: EVERY draw equals the reference's draw until one whose variate the reference itself
    recorded within bf16 noise of a step of the CDF it was drawn from; nothing is claimed after that draw."""
    z = load(golden_dir, "g6_decode.npz")
    model = _build(golden_dir, z, z[f"{tag}_bias"])
    temp, glen = float(z[f"{tag}_cfg"][0]), int(z[f"{tag}_cfg"][3])
    dec = _decoder(model, z, [tag])
    ref_draws, margins = z[f"{tag}_draw_tokens"].tolist(), z[f"{tag}_draw_margin"].tolist()
    draws = []
    with torch.no_grad():
        for _ in range(glen + 1):
            dec.iteration()
            if int(dec.draw[0]):
                draws.append(int(dec.token[0]))
            if int(dec.fsm[0, 5]):
                break
    seqs, traces = dec.sequences()
    got, ref = seqs[0], z[f"{tag}_seq"].tolist()
    ref_trace = [tuple(t) for t in z[f"{tag}_trace"].tolist()]
    if temp == 0:
        assert float(z[f"{tag}_min_gap"]) > 0.1          # (min top1 - top2 logit gap of the fixture)
        assert got == ref
        assert traces[0] == ref_trace
        return
    k = next((i for i in range(min(len(draws), len(ref_draws))) if draws[i] != ref_draws[i]), None)
    if tag == "sample8m":
        assert min(margins) > 0.04 and k is None
    if k is None:
        assert draws == ref_draws and got == ref and traces[0] == ref_trace
    else:
        # draws 0 .. k-1 agree (checked by the search); draw k may differ only if ITS variate sat on a CDF step
        assert margins[k] < 2e-2, (k, margins[k], draws[k], ref_draws[k])


@pytest.mark.parametrize("tags", [("greedy8", "greedy5"), ("sample8", "sample4x")])
def test_batched_kv_cache_decode_vs_reference_trace(golden_dir, tags):
    """Several sequences decoded IN PARALLEL by the captured hipGraph (K/V cache, ragged lengths, per-sequence
    forcing state) must each reproduce the reference's sequential trace."""
    from commu_amd.generate import BatchedGenerator
    z = load(golden_dir, "g6_decode.npz")
    model = _build(golden_dir, z, z[f"{tags[0]}_bias"])
    assert np.array_equal(z[f"{tags[0]}_bias"], z[f"{tags[1]}_bias"])
    temp, _, top_k, _ = z[f"{tags[0]}_cfg"]
    datas, srcs, glen = [], [], 0
    for tag in tags:
        glen = max(glen, int(z[f"{tag}_cfg"][3]))
        datas.append(_data(z, tag))
        it = iter(z[f"{tag}_uniforms"].tolist())
        srcs.append(lambda it=it: next(it, 0.5))
    gen = BatchedGenerator(model, torch.device(DEV), generation_length=glen)
    gen.uniform_sources = srcs
    gen.trace = [[] for _ in tags]
    meta = z["encoded_meta"].tolist()
    seqs, reports = gen.generate([meta] * len(tags), datas, float(temp), int(top_k))
    for b, tag in enumerate(tags):
        ref = z[f"{tag}_seq"].tolist()
        got = seqs[b]
        ref_trace = [tuple(t) for t in z[f"{tag}_trace"].tolist()]
        if float(temp) == 0:
            # the shorter fixture stops after its own generation_length iterations
            n = len(ref)
            assert got[:n] == ref, tag
            assert gen.trace[b][:len(ref_trace)] == ref_trace, tag
            if len(got) == len(ref):
                assert reports[b].n_chords - reports[b].consumed == int(z[f"{tag}_remnant"])
        else:
            n = min(len(got), len(ref))
            first = next((i for i in range(n) if got[i] != ref[i]), None)
            assert first is None or first > 12, (tag, first)
            if first is None and len(got) == len(ref):
                assert gen.trace[b] == ref_trace


def test_hipgraph_decode_64_sequences_token_exact(golden_dir):
    """BASELINE.json configs[3]: 64 sequences in parallel through the CAPTURED iteration graph.
    (1) graph replay == eager launches of the same kernels, bit for bit (token buffers, state records, logits);
    (2) every one of the 64 sequences (alternating greedy fixtures) is token-exact against the reference's own
        sequential generation, including the model-step trace with its discarded / doubled memory steps."""
    z = load(golden_dir, "g6_decode.npz")
    model = _build(golden_dir, z, z["greedy8_bias"])
    tags = ["greedy8", "greedy5", "greedy5", "greedy8"] * 16
    results = []
    for use_graph in (True, False):
        dec = _decoder(model, z, tags)
        with torch.no_grad():
            dec.run(use_graph=use_graph)
        torch.cuda.synchronize()
        results.append((dec.seq.clone(), dec.fsm.clone(), dec.state.logits.clone(), dec.state.klen.clone(),
                        dec.sequences()))
    (sg, fg, lg, kg, outg), (se, fe, le, ke, oute) = results
    assert torch.equal(sg, se) and torch.equal(fg, fe) and torch.equal(kg, ke)
    assert torch.equal(lg, le)
    seqs, traces = outg
    for b, tag in enumerate(tags):
        ref = z[f"{tag}_seq"].tolist()
        ref_trace = [tuple(t) for t in z[f"{tag}_trace"].tolist()]
        assert seqs[b][:len(ref)] == ref, (b, tag)
        assert traces[b][:len(ref_trace)] == ref_trace, (b, tag)


def test_sampled_hipgraph_matches_eager_and_margin_fixture(golden_dir):
    """Sampling inside the captured graph: 16 copies of the margin-enforced sampled fixture are reproduced token for
    token, and graph replay equals eager execution."""
    z = load(golden_dir, "g6_decode.npz")
    model = _build(golden_dir, z, z["sample8m_bias"])
    tags = ["sample8m"] * 16
    outs = []
    for use_graph in (True, False):
        dec = _decoder(model, z, tags)
        with torch.no_grad():
            dec.run(use_graph=use_graph)
        outs.append(dec.sequences())
    assert outs[0] == outs[1]
    ref = z["sample8m_seq"].tolist()
    ref_trace = [tuple(t) for t in z["sample8m_trace"].tolist()]
    for b in range(16):
        assert outs[0][0][b] == ref and outs[0][1][b] == ref_trace


def test_nucleus_mode_in_the_decode_loop(golden_dir):
    """top_p < 1 through the whole loop (the fused sampling / forcing launch inside the graph): replay equals eager
    execution, the sequences differ from the top-k-only run of the same variates (the filter is active), and every drawn
    token lies inside the nucleus of its step's distribution (checked with the per-stage launches on the first draws)."""
    z = load(golden_dir, "g6_decode.npz")
    model = _build(golden_dir, z, z["sample8m_bias"])
    tags = ["sample8m"] * 8
    outs = []
    for use_graph in (True, False):
        dec = _decoder(model, z, tags, top_p=0.5)
        with torch.no_grad():
            dec.run(use_graph=use_graph)
        outs.append(dec.sequences())
    assert outs[0] == outs[1]
    assert outs[0][0][0] != z["sample8m_seq"].tolist()
    dec = _decoder(model, z, tags, top_p=0.5)
    with torch.no_grad():
        for _ in range(24):
            dec.iteration(want_probs=True)
            torch.cuda.synchronize()
            drew, tok, pr = dec.draw.cpu(), dec.token.cpu(), dec.probs.cpu()
            for b in range(8):
                if drew[b] and tok[b] >= 0:
                    srt = torch.sort(pr[b].double(), descending=True).values
                    assert pr[b, tok[b]] > 0 and float(srt[srt > 0].sum()) > 0.999
                    # the kept set is a nucleus: dropping its least likely member leaves less than top_p of the ORIGINAL
                    # mass -- equivalently the kept set has at most as many members as the top-k set
                    assert int((pr[b] > 0).sum()) <= 32


def test_decoder_refuses_a_generation_longer_than_its_memory(golden_dir):
    """The K/V-cache step has no sliding memory window: asking for more iterations than cache rows is an error, not a
    silent overwrite of the last row (the reference would start dropping its oldest memory there)."""
    from commu_amd._lib import CommuHipError
    from commu_amd.generate import ForcedDecoder
    z = load(golden_dir, "g6_decode.npz")
    model = _build(golden_dir, z, z["greedy8_bias"])
    with pytest.raises(CommuHipError):
        ForcedDecoder(model, 2, generation_length=300, memory_length=200, temperature=0.0, top_k=32)


def test_token_generation_pipeline_from_reference_arguments(golden_dir):
    """generate.py's argument dictionary -> validated token sequences, 4 in parallel, with the fixture model whose
    output bias makes valid ComMU grammar likely (g6).  Every returned sequence passes the reference's validators
    and carries the forced chord progression."""
    from commu_amd.midi_generator.generate_pipeline import TokenGenerationPipeline
    from commu_amd.midi_generator.midi_inferrer import TOKEN_OFFSET
    z = load(golden_dir, "g6_decode.npz")
    model = _build(golden_dir, z, z["sample8_bias"])
    prog = "-".join(["Am"] * 8 + ["G"] * 8 + ["F"] * 8 + ["E"] * 8)
    args = dict(bpm=70, audio_key="aminor", time_signature="4/4", pitch_range="mid_high", num_measures=8.0,
                inst="acoustic_piano", genre="newage", min_velocity=60, max_velocity=80, track_role="main_melody",
                rhythm="standard", chord_progression=prog + "-" + prog, num_generate=4, top_k=32, temperature=0.95)
    pipe = TokenGenerationPipeline(model, torch.device(DEV), generation_length=400)
    seqs = pipe.execute(args, max_rounds=2)
    assert pipe.preprocess_task.execute(args) == z["encoded_meta"].tolist()
    assert pipe.attempts == len(seqs) + len(pipe.rejected) and pipe.attempts >= 4
    forced_ok = seqs + [s for why, s in pipe.rejected if why == "no_note"]
    assert forced_ok, [why for why, _ in pipe.rejected]          # the chord forcing itself must succeed
    for s in forced_ok:
        assert s[0] == 0 and s[1:12] == z["encoded_meta"].tolist() and s[-1] == TOKEN_OFFSET.EOS
        assert s.count(TOKEN_OFFSET.BAR) == 8
        chords = [t for t in s if TOKEN_OFFSET.CHORD_START <= t <= TOKEN_OFFSET.CHORD_END]
        assert chords == z["sample8_chord_token"].tolist()
    # the fixture model (random weights + a grammar bias) hardly ever emits a full note: those are the rejections
    from commu_amd.midi_generator.midi_inferrer import InferenceTask
    chk = InferenceTask(torch.device(DEV))
    assert all(chk.validate_generated_sequence(s) for s in seqs)
    assert not any(chk.validate_generated_sequence(s) for why, s in pipe.rejected if why == "no_note")



@pytest.mark.parametrize("shape", [(6, 8, 512, 1024), (6, 10, 500, 1000)], ids=["L6_D512_dh64", "L6_D500_dh50"])
def test_cached_decode_at_long_memory_vs_oracle(shape):
    """The K/V-cache decode step where the bench times it (bench.py `long_memory`): a REAL prefill of 1000 tokens
    (the training kernels, model.forward_generate's path), then 16 cached single-token steps, every step compared with
    oracle.xl_ref.forward_generate, which -- like the reference (model.py:606-628, midi_inferrer.py:199-207) --
    recomputes the whole memory each step.  Logits <= 2e-2 of their range; greedy tokens exact wherever the oracle's
    top-1 / top-2 gap exceeds the logit error bound (the gap and the number of such steps are reported); both runs are
    fed the ORACLE's greedy token so one flipped near-tie cannot derail the comparison."""
    from commu_amd.generate import DecodeState
    from oracle import xl_ref as X
    from test_configs_gpu import build
    L, H, D, DI = shape
    B, T0, NSTEP = 2, 1000, 16
    model, cfg, s, params = build(L, H, D, DI, 1, 4146, seed=41)
    model.eval()
    model.same_length = True
    model.reset_length(1, 4146)
    g = torch.Generator().manual_seed(19)
    ctx = torch.randint(2, 729, (T0, B), generator=g)
    with torch.no_grad():
        ref_logits, ref_mems = X.forward_generate(params, s, ctx, None, 4146, True)
    # the reference-API path at this length
    logits, mems = model.forward_generate(ctx.to(DEV), None)
    rng = float(ref_logits.abs().max())
    assert float((logits.float().cpu() - ref_logits).abs().max()) / rng < 2e-2
    assert tuple(mems.shape) == tuple(ref_mems.shape)
    # the cached path
    st = DecodeState(model, B, T0 + NSTEP + 8)
    st.prefill(ctx.to(DEV))
    ones = torch.ones(B, dtype=torch.uint8, device=DEV)
    tok = ref_logits[-1].argmax(-1)                                  # [B]
    gaps, checked, worst = [], 0, 0.0
    for step in range(NSTEP):
        with torch.no_grad():
            ref, ref_mems = X.forward_generate(params, s, tok[None], ref_mems, 4146, True)
        lg = st.step(tok.to(DEV), ones, ones)[:, :729].float().cpu()
        err = float((lg - ref[0]).abs().max())
        worst = max(worst, err / rng)
        assert err / rng < 2e-2, (step, err, rng)
        top2 = ref[0].topk(2, dim=-1).values
        for b in range(B):
            gap = float(top2[b, 0] - top2[b, 1])
            gaps.append(gap)
            if gap > 2.5 * err:
                checked += 1
                assert int(lg[b].argmax()) == int(ref[0, b].argmax()), (step, b, gap, err)
        tok = ref[0].argmax(-1)
    assert int(st.klen[0]) == T0 + NSTEP
    print(f"long-memory decode {shape}: worst logit error {worst:.2e} of range, min top1-top2 gap {min(gaps):.3f}, "
          f"{checked}/{len(gaps)} greedy tokens checked exact")
    assert checked >= len(gaps) // 2


@pytest.mark.parametrize("loaded", [False, True])
def test_split_key_decode_attention_matches_the_unsplit_kernel(loaded):
    """commu_decode_attn_split (the keys of a (sequence, head) pair over up to 8 workgroups, the last arriver combines the
    partial (max, sum, P.V) records inside the launch) against commu_decode_attn on the same caches: ragged memories from 3
    to 4100 keys (pairs with <= 512 keys take the unsplit path inside the split launch), inactive sequences, repeated
    launches (the counters must come back to zero), and -- `loaded` -- a second stream keeping the GPU busy so that the
    workgroups of a pair start unevenly (a stale or missing record would be an O(1) error)."""
    from commu_amd._lib import call
    from commu_amd.ops import _p, _s
    B, H, DH, Lmax = 12, 8, 64, 4200
    g = torch.Generator().manual_seed(23)
    qkv = (torch.randn(B, 3 * H * DH, generator=g) * 0.7).to(torch.bfloat16).to(DEV)
    kc = (torch.randn(B, H, Lmax, DH, generator=g) * 0.7).to(torch.bfloat16).to(DEV)
    vc = (torch.randn(B, H, Lmax, DH, generator=g)).to(torch.bfloat16).to(DEV)
    rd = (torch.randn(Lmax, H * DH, generator=g) * 0.7).to(torch.bfloat16).to(DEV)
    u, vb = (torch.randn(H * DH, generator=g) * 0.3).to(DEV), (torch.randn(H * DH, generator=g) * 0.3).to(DEV)
    klen = torch.tensor([3, 511, 512, 513, 1000, 1024, 1500, 2047, 3000, 4100, 777, 4199 - 1], dtype=torch.int32, device=DEV)
    active = torch.ones(B, dtype=torch.uint8, device=DEV)
    active[10] = 0
    ws = torch.empty(B * H * 16 * (DH + 2), device=DEV, dtype=torch.float32).fill_(float("nan"))
    cnt = torch.zeros(B * H, device=DEV, dtype=torch.int32)
    ref = torch.zeros(B, H * DH, device=DEV, dtype=torch.bfloat16)
    call("commu_decode_attn", _p(qkv), qkv.stride(0), _p(kc), _p(vc), _p(rd), rd.stride(0), _p(u), _p(vb), _p(klen),
         _p(active), _p(ref), ref.stride(0), B, H, DH, Lmax, 0.125, 0, _s())
    side = torch.cuda.Stream()
    a_ = torch.randn(4096, 4096, device=DEV)
    for nsplit in (8, 3, 16):
        for rep in range(4):
            if loaded:
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    for _ in range(3):
                        a_ @ a_
            out = torch.zeros(B, H * DH, device=DEV, dtype=torch.bfloat16)
            call("commu_decode_attn_split", _p(qkv), qkv.stride(0), _p(kc), _p(vc), _p(rd), rd.stride(0), _p(u), _p(vb),
                 _p(klen), _p(active), _p(out), out.stride(0), B, H, DH, Lmax, 0.125, 0, nsplit, _p(ws), _p(cnt), _s())
            torch.cuda.synchronize()
            assert int(cnt.abs().sum()) == 0, (nsplit, rep)
            d = (out.float() - ref.float()).abs().max()
            assert float(d) < 2e-2 * float(ref.float().abs().max()), (nsplit, rep, float(d))
            assert float(out[10].float().abs().max()) == 0          # the inactive sequence is not touched


@pytest.mark.parametrize("B,loaded,shape", [(64, False, (6, 8, 512, 1024)), (64, True, (6, 8, 512, 1024)),
                                            (37, False, (6, 8, 512, 1024)), (3, True, (6, 8, 512, 1024)),
                                            (64, True, (6, 10, 500, 1000)), (5, False, (6, 10, 500, 1000)),
                                            (64, True, (3, 16, 1024, 2048)), (21, False, (3, 16, 1024, 2048))])
def test_layer_tail_launch_matches_the_per_linear_chain(B, loaded, shape):
    """commu_decode_layer_tail (four Linears + two LayerNorms of a layer as phases of one launch, hand-offs between
    workgroups inside the launch) against the chain of per-Linear launches it replaces, on the same model, caches and
    tokens (the last two shapes: d_model 1024 / d_inner 2048 / 16 heads, decode_tail_wide_kernel), for 160 consecutive steps:
    every step's logits within bf16 rounding of each other (a stale or torn hand-off
    would be an O(1) error in some row), the caches equal at the end, no workgroup gave up.  `loaded`: a second stream
    keeps the GPU busy with large GEMMs meanwhile, so the workgroups of a launch start and run unevenly."""
    import commu_amd.generate as G
    from test_configs_gpu import build
    model, cfg, s, params = build(*shape, 1, 4146, seed=23)          # (the second shape: the released default, zero-padded)
    model.eval()
    model.same_length = True
    model.reset_length(1, 4146)
    g = torch.Generator().manual_seed(5 + B)
    ctx = torch.randint(2, 729, (11, B), generator=g).to(DEV)
    NSTEP = 160
    toks = torch.randint(2, 729, (NSTEP, B), generator=g).to(DEV)
    acts = (torch.rand(NSTEP, B, generator=g) < 0.9).to(torch.uint8).to(DEV)
    st_a, st_b = G.DecodeState(model, B, 11 + NSTEP + 8), G.DecodeState(model, B, 11 + NSTEP + 8)
    assert st_a.tail_ok
    st_a.prefill(ctx)
    st_b.prefill(ctx)
    side = torch.cuda.Stream()
    big = torch.randn(4096, 4096, device=DEV, dtype=torch.bfloat16)
    worst = 0.0
    for i in range(NSTEP):
        if loaded and i % 4 == 0:
            with torch.cuda.stream(side):
                for _ in range(3):
                    big @ big
        assert G.USE_LAYER_TAIL
        la = st_a.step(toks[i], acts[i], acts[i]).clone()
        G.USE_LAYER_TAIL = False
        try:
            lb = st_b.step(toks[i], acts[i], acts[i]).clone()
        finally:
            G.USE_LAYER_TAIL = True
        rng = float(lb[:, :729].abs().max())
        err = float((la[:, :729] - lb[:, :729]).abs().max()) / rng
        worst = max(worst, err)
        assert err < 1e-2, (i, err)
    torch.cuda.synchronize()
    st_a.check()
    assert torch.equal(st_a.klen, st_b.klen)
    kerr = float((st_a.kc.float() - st_b.kc.float()).abs().max()) / float(st_b.kc.float().abs().max())
    verr = float((st_a.vc.float() - st_b.vc.float()).abs().max()) / float(st_b.vc.float().abs().max())
    print(f"layer tail vs per-Linear chain, {shape}, B={B}, loaded={loaded}: worst logit difference {worst:.2e} of range, "
          f"K cache {kerr:.2e}, V cache {verr:.2e}")
    assert kerr < 2e-2 and verr < 2e-2


def test_decoder_reused_after_a_weight_update():
    """A ForcedDecoder (and its captured graph) outlives weight updates: load() refreshes, IN PLACE, everything the decode
    step derives from the weights -- the packed copies the layer-tail launches read and the r_net distance tables -- so the
    replayed graph decodes with the new weights.  Sampled sequences (same variates) of the reused decoder == a fresh
    decoder's (eager), and != the sequences before the update."""
    from commu_amd.generate import ForcedDecoder
    from test_configs_gpu import build
    model, cfg, s, params = build(6, 8, 512, 1024, 1, 4146, seed=31)
    model.eval()
    model.same_length = True
    model.reset_length(1, 4146)
    with torch.no_grad():
        bias = model.crit.out_layers[0].bias
        bias.zero_()
        bias[1:3] = -1e9                      # no EOS / BAR, no chord tokens: every iteration is a model step and a draw
        bias[195:304] = -1e9
    meta = [574, 623, 627, 635, 639, 642, 651, 684, 694, 720, 727]
    data = types.SimpleNamespace(num_measures=4.0, chord_token_components={"chord_token": [], "chord_position": []})

    uni = np.random.RandomState(4).random_sample((4, 64)).astype(np.float32)

    def run(dec, graph):
        dec.load([meta] * 4, [data] * 4, uni)
        with torch.no_grad():
            dec.run(use_graph=graph)
        return dec.sequences()[0]
    dec = ForcedDecoder(model, 4, generation_length=40, memory_length=4146, temperature=0.95, top_k=32)
    assert dec.state.tail_ok
    before = run(dec, True)
    g = torch.Generator().manual_seed(3)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if p.dim() == 2 and "emb" not in n:
                p.add_((0.05 * torch.randn(p.shape, generator=g)).to(p.device))
    model._refresh_shadows()
    after = run(dec, True)                                   # same decoder, same captured graph
    fresh = run(ForcedDecoder(model, 4, generation_length=40, memory_length=4146, temperature=0.95, top_k=32), False)
    assert after == fresh
    assert after != before


def test_wide_model_decoder_graph_and_layer_tail():
    """The decode loop of a wide model (d_model 1024, d_inner 2048, 16 heads: the width of BASELINE.json configs[4]) through
    decode_tail_wide_kernel: the captured iteration graph replays the eager launches token for token, and the sampled
    sequences (same variates) equal those of the per-Linear chain."""
    import commu_amd.generate as G
    from commu_amd.generate import ForcedDecoder
    from test_configs_gpu import build
    model, cfg, s, params = build(3, 16, 1024, 2048, 1, 4146, seed=41)
    model.eval()
    model.same_length = True
    model.reset_length(1, 4146)
    with torch.no_grad():
        bias = model.crit.out_layers[0].bias
        bias.zero_()
        bias[1:3] = -1e9
        bias[195:304] = -1e9
    meta = [574, 623, 627, 635, 639, 642, 651, 684, 694, 720, 727]
    data = types.SimpleNamespace(num_measures=4.0, chord_token_components={"chord_token": [], "chord_position": []})
    uni = np.random.RandomState(9).random_sample((6, 64)).astype(np.float32)

    def run(graph):
        dec = ForcedDecoder(model, 6, generation_length=48, memory_length=4146, temperature=0.95, top_k=32)
        dec.load([meta] * 6, [data] * 6, uni)
        with torch.no_grad():
            dec.run(use_graph=graph)
        dec.state.check()
        return dec, dec.sequences()[0]
    dec_g, seq_g = run(True)
    assert dec_g.state.tail_ok
    _, seq_e = run(False)
    assert seq_g == seq_e
    G.USE_LAYER_TAIL = False
    try:
        _, seq_chain = run(False)
    finally:
        G.USE_LAYER_TAIL = True
    # (bf16 rounding differs between the two step implementations: a draw that sits on a probability boundary may differ;
    #  the sequences must agree on almost every token and exactly up to the first such draw)
    same = sum(int(a == b) for sa, sb in zip(seq_g, seq_chain) for a, b in zip(sa, sb))
    total = sum(min(len(sa), len(sb)) for sa, sb in zip(seq_g, seq_chain))
    assert total > 0 and same >= 0.9 * total, (same, total)


def test_generate_stream_rearms_finished_slots():
    """BatchedGenerator.generate_stream: 150 attempts of one request through 64 slots, a finished slot re-armed with the
    next attempt while the others keep decoding.  Every attempt draws from its own variate stream, so its sequence must
    be the one the attempt produces when it is decoded in a plain 64-way batch from a fresh start (a re-armed slot reuses
    the context rows of its K/V cache and resets everything else): attempts 0..63 (first occupants) and 64..127 (second
    occupants of the same slots) are compared token for token."""
    from commu_amd.generate import BatchedGenerator, ForcedDecoder
    from test_configs_gpu import build
    model, cfg, s, params = build(6, 8, 512, 1024, 1, 4146, seed=37)
    model.eval()
    model.same_length = True
    model.reset_length(1, 4146)
    with torch.no_grad():
        bias = model.crit.out_layers[0].bias
        bias.zero_()
        bias[1:3] = -1e9                      # no EOS / BAR, no chord tokens: every attempt runs its 48 iterations
        bias[195:304] = -1e9
    meta = [574, 623, 627, 635, 639, 642, 651, 684, 694, 720, 727]
    data = types.SimpleNamespace(num_measures=4.0, chord_token_components={"chord_token": [], "chord_position": []})
    gen = BatchedGenerator(model, torch.device(DEV), generation_length=48, memory_length=4146)
    seen = []

    def accept(seq, rep):
        seen.append(seq)
        return True
    out, started = gen.generate_stream(meta, data, 0.95, 32, need=150, accept=accept, slots=64, seed=11)
    assert len(out) == 150 and 150 <= started <= 214 and len(seen) >= 150
    assert all(s_ is not None and s_[:12] == [0] + meta and len(s_) == 12 + 48 for s_ in out)
    for first in (0, 64):
        dec = ForcedDecoder(model, 64, generation_length=48, memory_length=4146, temperature=0.95, top_k=32)
        uni = np.stack([BatchedGenerator.attempt_uniforms(11, first + b, dec.ld_u) for b in range(64)])
        dec.load([meta] * 64, [data] * 64, uni)
        with torch.no_grad():
            dec.run(use_graph=False)
        ref = dec.sequences()[0]
        assert out[first:first + 64] == ref, first
    assert len({tuple(s_) for s_ in out}) > 20           # the attempts differ (a random-init tied embedding mostly repeats its input)


def test_generate_stream_returns_the_sequential_loops_answer():
    """With ragged lengths (EOS is likely) and a validator that rejects some attempts, generate_stream returns the first
    `need` accepted attempts IN ATTEMPT ORDER -- what the reference's one-after-the-other loop (midi_inferrer.py:338-354)
    would return for the same per-attempt variates -- not the first to finish: compared with all attempts decoded in plain
    64-way batches and filtered in order.  Slots are re-armed at different times here."""
    from commu_amd.generate import BatchedGenerator, ForcedDecoder
    from test_configs_gpu import build
    model, cfg, s, params = build(6, 8, 512, 1024, 1, 4146, seed=41)
    model.eval()
    model.same_length = True
    model.reset_length(1, 4146)
    with torch.no_grad():
        bias = model.crit.out_layers[0].bias
        bias.zero_()
        bias[1] = 3.0                         # EOS is drawn every few dozen tokens: lengths differ
        bias[2] = -1e9
        bias[195:304] = -1e9
    meta = [574, 623, 627, 635, 639, 642, 651, 684, 694, 720, 727]
    data = types.SimpleNamespace(num_measures=4.0, chord_token_components={"chord_token": [], "chord_position": []})
    GL, NEED = 96, 50

    def accept(seq, rep):
        return seq is not None and len(seq) % 3 != 0
    gen = BatchedGenerator(model, torch.device(DEV), generation_length=GL, memory_length=4146)
    out, started = gen.generate_stream(meta, data, 0.95, 32, need=NEED, accept=accept, slots=64, seed=5)
    allseq = []
    for first in range(0, 256, 64):
        dec = ForcedDecoder(model, 64, generation_length=GL, memory_length=4146, temperature=0.95, top_k=32)
        uni = np.stack([BatchedGenerator.attempt_uniforms(5, first + b, dec.ld_u) for b in range(64)])
        dec.load([meta] * 64, [data] * 64, uni)
        with torch.no_grad():
            dec.run(use_graph=False)
        allseq += dec.sequences()[0]
    lens = sorted(len(s_) for s_ in allseq)
    assert lens[0] < lens[-1] - 20, "the attempts should have different lengths"
    want = [s_ for s_ in allseq if accept(s_, None)][:NEED]
    assert len(want) == NEED and out == want
    assert NEED <= started <= 256
