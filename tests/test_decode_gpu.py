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


@pytest.mark.parametrize("tag", ["greedy8", "greedy5", "sample8", "sample4x"])
def test_decode_loop_vs_reference_trace(golden_dir, tag):
    from commu_amd.midi_generator.midi_inferrer import InferenceTask
    z = load(golden_dir, "g6_decode.npz")
    model = _build(golden_dir, z, z[f"{tag}_bias"])
    temp, nm, top_k, glen = z[f"{tag}_cfg"]
    input_data = types.SimpleNamespace(
        temperature=float(temp), top_k=int(top_k), num_generate=1, num_measures=float(nm),
        chord_token_components={"chord_token": z[f"{tag}_chord_token"].tolist(),
                                "chord_position": z[f"{tag}_chord_position"].tolist()})
    task = InferenceTask(torch.device(DEV))
    task(model=model, input_data=input_data,
         inference_cfg=types.SimpleNamespace(GENERATION=types.SimpleNamespace(generation_length=int(glen))))
    us = iter(z[f"{tag}_uniforms"].tolist())
    drawn = []                                   # (u, token, probs) per draw

    def src():
        u = next(us, 0.5)                        # only consumed past a divergence
        drawn.append([u])
        return u
    task.uniform_source = src
    orig_sample = task.sample

    def sample(logits_row, wrong_tokens):
        t, probs = orig_sample(logits_row, wrong_tokens, want_probs=True)
        if drawn and len(drawn[-1]) == 1:
            drawn[-1] += [t, probs[0].double().cpu()]
        return t
    task.sample = sample
    task.trace = []
    meta = z["encoded_meta"].tolist()
    with torch.no_grad():
        seq, mems = task.init_seq_and_mems(meta, len(meta))
        task.generate_sequence(seq, mems)
    got, ref = task.last_raw_seq, z[f"{tag}_seq"].tolist()
    ref_trace = z[f"{tag}_trace"].tolist()
    if float(temp) == 0:
        # greedy: token-exact (min top1-top2 logit gap of the fixture is reported in the fixture)
        assert float(z[f"{tag}_min_gap"]) > 0.1
        assert got == ref
        assert [list(t) for t in task.trace] == ref_trace
    else:
        # sampled: identical until (possibly) a draw whose variate sits within bf16 noise of a CDF step
        n = min(len(got), len(ref))
        first = next((i for i in range(n) if got[i] != ref[i]), None)
        if first is None:
            assert got == ref and [list(t) for t in task.trace] == ref_trace
        else:
            # the first differing token must come from a draw whose variate lies within bf16 noise of a
            # step of the CDF it was drawn from (every later difference is a consequence of it)
            near = [float((torch.cumsum(p, 0) - u).abs().min()) for u, t, p in (d for d in drawn if len(d) == 3)]
            assert min(near) < 2e-2, (first, min(near))
            assert first > 12


@pytest.mark.parametrize("tags", [("greedy8", "greedy5"), ("sample8", "sample4x")])
def test_batched_kv_cache_decode_vs_reference_trace(golden_dir, tags):
    """Several sequences decoded IN PARALLEL (K/V cache, ragged lengths, per-sequence forcing state) must each
    reproduce the reference's sequential trace."""
    from commu_amd.generate import BatchedGenerator
    z = load(golden_dir, "g6_decode.npz")
    model = _build(golden_dir, z, z[f"{tags[0]}_bias"])
    assert np.array_equal(z[f"{tags[0]}_bias"], z[f"{tags[1]}_bias"])
    temp, _, top_k, _ = z[f"{tags[0]}_cfg"]
    datas, srcs, glen = [], [], 0
    for tag in tags:
        t, nm, k, gl = z[f"{tag}_cfg"]
        assert t == temp
        glen = max(glen, int(gl))
        datas.append(types.SimpleNamespace(num_measures=float(nm), chord_token_components={
            "chord_token": z[f"{tag}_chord_token"].tolist(), "chord_position": z[f"{tag}_chord_position"].tolist()}))
        it = iter(z[f"{tag}_uniforms"].tolist())
        srcs.append(lambda it=it: next(it, 0.5))
    gen = BatchedGenerator(model, torch.device(DEV), generation_length=glen)
    gen.uniform_sources = srcs
    gen.trace = [[] for _ in tags]
    meta = z["encoded_meta"].tolist()
    seqs, teachers = gen.generate([meta] * len(tags), datas, float(temp), int(top_k))
    for b, tag in enumerate(tags):
        ref = z[f"{tag}_seq"].tolist()
        gl = int(z[f"{tag}_cfg"][3])
        got = seqs[b]
        ref_trace = [tuple(t) for t in z[f"{tag}_trace"].tolist()]
        if float(temp) == 0:
            # the shorter fixture stops after its own generation_length iterations
            n = len(ref)
            assert got[:n] == ref, tag
            assert gen.trace[b][:len(ref_trace)] == ref_trace, tag
        else:
            n = min(len(got), len(ref))
            first = next((i for i in range(n) if got[i] != ref[i]), None)
            assert first is None or first > 12, (tag, first)
            if first is None and len(got) == len(ref):
                assert gen.trace[b] == ref_trace


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
