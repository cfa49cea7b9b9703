"""CPU tests of the host-side mirror of the reference interface (no GPU, no kernels):
batch producer vs the reference's iterators (fixture g8_dataset), LR schedule, config surface,
state_dict names."""
import os

import numpy as np
import pytest
import torch

from commu_amd.model.config_helper import get_cfg, get_default_cfg_inference, get_default_cfg_training
from commu_amd.model.dataset import BaseVocab, ComMUDataset, synthetic_batch
from commu_amd.optim import lr_lambda_factory


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def corpus_from_fixture(z):
    out = {}
    for split, key in (("train", "train"), ("valid", "val")):
        lens = z[f"{key}_lens"]
        ev = z[f"{key}_events"]
        metas = z[f"{key}_meta"]
        seqs, o = [], 0
        for i, n in enumerate(lens):
            seqs.append(np.concatenate([metas[i], ev[o:o + n]]))
            o += n
        out[split] = seqs
    return out


def test_train_iterator_matches_reference(golden_dir):
    z = load(golden_dir, "g8_dataset.npz")
    ds = ComMUDataset(None, None, sequences=corpus_from_fixture(z))
    it = ds.get_iterator(4, 16, "cpu", "train", True, seed=1111)()
    for i in range(int(z["tr_n"])):
        data, target, reset, ntok = next(it)
        assert np.array_equal(data.numpy(), z[f"tr{i}_data"]), i
        assert np.array_equal(target.numpy(), z[f"tr{i}_target"]), i
        assert np.array_equal(reset.numpy(), z[f"tr{i}_reset"]), i
        assert ntok == int(z[f"tr{i}_ntok"])
    batches = list(ds.get_iterator(3, 8, "cpu", "valid", False, seed=None)())
    assert len(batches) == int(z["noshuf_n"])
    for i, (data, target, reset, ntok) in enumerate(batches):
        assert np.array_equal(data.numpy(), z[f"ns{i}_data"]) and np.array_equal(target.numpy(), z[f"ns{i}_target"])
        assert np.array_equal(reset.numpy(), z[f"ns{i}_reset"]) and ntok == int(z[f"ns{i}_ntok"])


def test_eval_iterator_rank_shards_match_reference(golden_dir):
    z = load(golden_dir, "g8_dataset.npz")
    ds = ComMUDataset(None, None, sequences=corpus_from_fixture(z))
    for ws in (1, 2):
        for rank in range(ws):
            evs = list(ds.eval_iterator(4, 16, "cpu", "valid", local_rank=rank, world_size=ws)())
            assert len(evs) == int(z[f"ev_ws{ws}_r{rank}_n"])
            for i, (data, target, allreset, ntok) in enumerate(evs):
                pre = f"ev_ws{ws}_r{rank}_{i}_"
                assert np.array_equal(data.numpy(), z[pre + "data"]) and np.array_equal(target.numpy(), z[pre + "target"])
                assert bool(allreset) == bool(z[pre + "reset"]) and ntok == int(z[pre + "ntok"])


def test_npy_on_disk_format_roundtrip(golden_dir, tmp_path):
    z = load(golden_dir, "g8_dataset.npz")
    seqs = corpus_from_fixture(z)
    for split, tag in (("train", "train"), ("valid", "val")):
        metas = np.array([np.array(s[:11], dtype=object) for s in seqs[split]], dtype=object)
        ev = np.empty(len(seqs[split]), dtype=object)
        for i, s in enumerate(seqs[split]):
            ev[i] = s[11:].astype(np.int16)
        np.save(tmp_path / f"input_{tag}.npy", metas, allow_pickle=True)
        np.save(tmp_path / f"target_{tag}.npy", ev, allow_pickle=True)
    ds = ComMUDataset(str(tmp_path), None)
    data, target, reset, ntok = next(ds.get_iterator(4, 16, "cpu", "train", True, seed=1111)())
    assert np.array_equal(data.numpy(), z["tr0_data"]) and ntok == int(z["tr0_ntok"])
    assert len(ds.vocab) == 729 and ds.vocab.pad_id == 0


def test_lr_schedule_matches_reference(golden_dir):
    z = load(golden_dir, "g8_optim.npz")
    f = lr_lambda_factory(100, 0.004, 0.0001)
    assert np.array_equal(np.array([f(s) for s in range(301)]), z["lr_lambda_0_300"])
    assert np.array_equal(np.array([f(s) for s in (1000, 10000, 20000, 200000)]), z["lr_lambda_far"])
    assert f(0) == 0.0 and lr_lambda_factory(0, 0.004, 0.0001)(0) == 1.0


def test_config_surface():
    cfg = get_default_cfg_training()
    assert (cfg.MODEL.num_layers, cfg.MODEL.num_heads, cfg.MODEL.units, cfg.MODEL.inner_size) == (6, 10, 500, 1000)
    assert (cfg.TRAIN.batch_size, cfg.TRAIN.batch_chunk, cfg.TRAIN.tgt_length, cfg.TRAIN.mem_length) == (256, 4, 128, 1024)
    assert (cfg.TRAIN.lr, cfg.TRAIN.lr_min, cfg.TRAIN.warmup_step, cfg.TRAIN.clip) == (0.004, 0.0001, 100, 1.0)
    assert (cfg.EVALUATE.batch_size, cfg.EVALUATE.tgt_length, cfg.EVALUATE.mem_length) == (10, 128, 2048)
    with pytest.raises(AttributeError):
        cfg.MODEL.units = 1
    cfg.defrost()
    cfg.MODEL.same_length = True
    cfg.freeze()
    inf = get_default_cfg_inference()
    assert inf.MODEL.memory_length == 4146 and inf.GENERATION.generation_length == 4096
    assert inf.SAMPLING.temperature == 0.95 and inf.SAMPLING.threshold == 32.0
    b = get_cfg()
    assert (b.MODEL.units, b.MODEL.num_heads, b.MODEL.inner_size, b.TRAIN.tgt_length) == (512, 8, 1024, 1024)


def test_model_state_dict_names(golden_dir):
    from commu_amd.model.model import MemTransformerLM
    z = load(golden_dir, "g1_train_mem.npz")
    L, H, D, DI, T, B, mem_len, sl = [int(x) for x in z["meta"][:8]]
    m = MemTransformerLM(get_cfg(num_layers=L, num_heads=H, units=D, inner_size=DI, tgt_length=T, mem_length=mem_len),
                         BaseVocab())
    sd = m.state_dict()
    ref = {k[3:]: z[k].shape for k in z.files if k.startswith("p::")}
    assert set(sd) == set(ref) | {"crit.out_layers.0.weight"}
    for k, shp in ref.items():
        assert tuple(sd[k].shape) == shp, k
    assert m.crit.out_layers[0].weight is m.word_emb.emb_layers[0].weight     # model.py:480-481
    assert m.crit.n_clusters == 0
    assert m.init_mems(L).shape == (L + 1, 0)
    m.reset_length(1, 0)
    assert m.init_mems(L) is None


def test_synthetic_batch_shape():
    d, t, r, n = synthetic_batch(16, 4, "cpu", seed=1)
    assert d.shape == (16, 4) and n == 64 and int(d.min()) >= 2 and int(d.max()) < 729
    assert torch.equal(d[1:], t[:-1]) and not bool(r.any())


def test_meta_and_chord_encoding_vs_reference_cases(golden_dir):
    """commu_amd.midi_generator.meta against 260 seeded inputs run through the reference's PreprocessTask
    (tests/golden/make_golden.py g7c): tokens, chord components, and WHICH inputs are rejected."""
    import gzip
    import json
    from commu_amd.midi_generator import meta as MM
    with gzip.open(os.path.join(golden_dir, "g7_meta_cases.json.gz"), "rt") as f:
        cases = json.load(f)
    n_meta = n_chord = 0
    for c in cases:
        args = c["args"]
        try:
            data = MM.InputData(**args)
        except (ValueError, TypeError):
            assert c.get("meta_error") == "ValidationError" or c.get("chord_error") == "ValidationError", args
            continue
        if c.get("meta_error") == "ValidationError":
            # pydantic type validation of the reference (e.g. "unknown" in an int field): not a token-level case
            continue
        if "encoded_meta" in c:
            assert MM.encode_meta(data) == c["encoded_meta"], args
            n_meta += 1
        else:
            with pytest.raises(MM.UnprocessableMidiError):
                MM.encode_meta(data)
        if "chord_token" in c:
            comp = data.chord_token_components
            assert comp["chord_token"] == c["chord_token"] and comp["chord_position"] == c["chord_position"], args
            n_chord += 1
        elif c.get("chord_error") == "KeyError":
            with pytest.raises(KeyError):
                data.chord_token_components
    assert n_meta > 150 and n_chord > 100


def test_meta_readme_example(golden_dir):
    from commu_amd.midi_generator.meta import PreprocessTask
    z = np.load(os.path.join(golden_dir, "g7_meta.npz"))
    prog = "-".join(["Am"] * 8 + ["G"] * 8 + ["F"] * 8 + ["E"] * 8) * 1
    prog = prog + "-" + prog
    t = PreprocessTask()
    enc = t.execute(dict(bpm=70, audio_key="aminor", time_signature="4/4", pitch_range="mid_high", num_measures=8.0,
                         inst="acoustic_piano", genre="newage", min_velocity=60, max_velocity=80,
                         track_role="main_melody", rhythm="standard", chord_progression=prog, num_generate=1,
                         top_k=32, temperature=0.95))
    assert enc == z["encoded_meta"].tolist()
    comp = t.input_data.chord_token_components
    assert comp["chord_token"] == z["chord_token"].tolist() and comp["chord_position"] == z["chord_position"].tolist()


def test_note_validator_matches_reference_scan():
    """validate_generated_sequence (midi_inferrer.py:322-336): a note = POSITION, VELOCITY, PITCH, DURATION in a row
    (the scan visits every index that still has two successors); ForcingReport restates the structural check of
    validate_teacher_forced_sequence (:146-169)."""
    from commu_amd.midi_generator.midi_inferrer import ForcingReport, count_notes
    note = [440, 150, 60, 320]
    assert count_notes([0, 2] + note + [1]) == 1
    assert count_notes([0, 2] + note) == 1
    assert count_notes([0, 2, 440, 150, 60]) == 0            # no duration token
    assert count_notes([0, 2, 440, 150, 60, 2, 1]) == 0
    assert count_notes([0] + note + note + [1]) == 2
    rep = ForcingReport(4, 4.0)
    assert rep.length_fit
    seq = [0] + [2, 432, 199] * 4 + [1]
    rep.consumed = 4
    rep.validate_teacher_forced_sequence(seq)
    rep.consumed = 3
    with pytest.raises(Exception, match="remnant chord"):
        rep.validate_teacher_forced_sequence(seq)
    rep.consumed = 4
    with pytest.raises(Exception, match="bar length"):
        rep.validate_teacher_forced_sequence(seq[:-4] + [1])


@pytest.mark.parametrize("seed,nseq,B,T", [(0, 23, 4, 8), (1, 9, 8, 5), (2, 60, 7, 16), (3, 12, 3, 4)])
def test_epoch_scheduler_matches_the_reference_loop_over_epochs(seed, nseq, B, T):
    """The list-scheduling epoch planner + vectorised gather against the oracle's restatement of the reference's
    per-column / per-batch loop (dataset.py:117-183): ragged corpora with one-token sequences, columns that run dry,
    three shuffled epochs and the single unshuffled pass."""
    from oracle.dataset_ref import packed_batches
    rng = np.random.RandomState(seed)
    lens = rng.randint(0, 4 * T, size=nseq)
    lens[rng.randint(0, nseq, size=3)] = 0                     # sequences that are only the start token
    raw = [rng.randint(1, 729, size=n) for n in lens]
    ds = ComMUDataset(None, None, sequences={"train": raw, "valid": raw})
    seqs = [np.insert(a.astype(np.int64), 0, 0) for a in raw]
    for shuffle in (True, False):
        want = list(packed_batches(seqs, B, T, shuffle, 77 if shuffle else None, max_batches=10 ** 6 if not shuffle else 0))
        if shuffle:
            per_epoch = sum(1 for _ in packed_batches(seqs, B, T, False, None, 10 ** 6))
            want = list(packed_batches(seqs, B, T, True, 77, max_batches=3 * per_epoch + 2))
        it = ds.get_iterator(B, T, "cpu", "train", shuffle, seed=77 if shuffle else None)()
        got = [next(it) for _ in range(len(want))] if shuffle else list(it)
        assert len(got) == len(want)
        for i, ((d, t, r, n), (wd, wt, wr, wn)) in enumerate(zip(got, want)):
            assert np.array_equal(d.numpy(), wd) and np.array_equal(t.numpy(), wt), (shuffle, i)
            assert np.array_equal(r.numpy(), wr) and n == wn, (shuffle, i)


def test_weights_init_matches_reference_statistics(golden_dir):
    """weights_init (train.py:291-342) as restated in commu_amd.train: per-tensor moments against the ones the
    reference's own function produced on the same shape (g10 fixture): N(0, 0.01) Linear / Embedding weights and
    r_*_bias, zero biases, N(1, 0.01) LayerNorm weights."""
    from commu_amd.train import build_model
    z = load(golden_dir, "g10_checkpoint.npz")
    cfg = get_cfg(num_layers=2, num_heads=4, units=256, inner_size=512, tgt_length=16, mem_length=16)
    model = build_model(cfg, BaseVocab(), "cpu", seed=3)
    sd = model.state_dict()
    keys = [k[6:] for k in z.files if k.startswith("init::")]
    assert set(keys) == {k for k in sd if k not in ("crit.out_layers.0.weight", "pos_emb.inv_freq")}
    for k in keys:
        mean_ref, std_ref, amax_ref = z["init::" + k]
        v = sd[k].float()
        if amax_ref == 0:                                   # biases
            assert float(v.abs().max()) == 0, k
            continue
        n = v.numel()
        assert abs(float(v.mean()) - mean_ref) < 6 * std_ref / np.sqrt(n) + 1e-6, k
        assert abs(float(v.std()) / std_ref - 1) < 0.25, k
    assert abs(float(sd["layers.0.pos_ff.layer_norm.weight"].mean()) - 1.0) < 5e-3


def test_reference_written_checkpoint_is_readable_without_the_reference(golden_dir):
    """train.py:39-48 layout incl. the pickled `commu.model.dataset.BaseVocab` instance."""
    from commu_amd.train import read_checkpoint
    ck = read_checkpoint(os.path.join(golden_dir, "g10_checkpoint.pt"))
    assert sorted(ck) == ["amp", "best_val_loss", "model", "optimizer", "scheduler", "train_step", "vocab"]
    assert isinstance(ck["vocab"], BaseVocab) and len(ck["vocab"]) == 729 and ck["amp"] is None
    assert ck["train_step"] == 3 and "layers.1.dec_attn.r_net.weight" in ck["model"]
    assert set(ck["optimizer"]["state"][0]) >= {"step", "exp_avg", "exp_avg_sq"}


def test_abandoned_iterator_releases_its_prefetch_thread():
    """A train iterator dropped after one batch (eval iterators are recreated every eval interval; generators are
    dropped early): the prefetch worker -- possibly blocked on its full ready queue -- must terminate."""
    import threading
    rng = np.random.default_rng(3)
    seqs = [rng.integers(2, 700, size=int(n)).astype(np.int64) for n in rng.integers(40, 90, size=24)]
    cfg = get_cfg(tgt_length=16, mem_length=0, batch_size=4)
    ds = ComMUDataset(None, cfg, sequences={"train": seqs, "valid": seqs[:6]})
    before = threading.active_count()
    for _ in range(3):
        it = ds.get_iterator(4, 16, "cpu", "train", True, seed=1)()
        next(it)
        import time
        time.sleep(0.05)                                          # let the worker fill its queue and block
        it.close()
    deadline = time.time() + 5.0
    while threading.active_count() > before and time.time() < deadline:
        time.sleep(0.05)
    assert threading.active_count() <= before


@pytest.mark.parametrize("over,frag", [
    (dict(units=1024, num_heads=8), "d_head <= 64"),          # d_head 128
    (dict(units=2048, num_heads=32), "d_model <= 1024"),
    (dict(units=130, num_heads=2), "d_model % 4 == 0"),
])
def test_unsupported_shapes_raise_with_the_supported_list(over, frag):
    """Shapes the reference accepts (model.py:429-444 has no limits) and this build does not: refused at construction
    with an error that lists what IS supported ("Supported shapes")."""
    from commu_amd._lib import CommuHipError
    from commu_amd.model.model import MemTransformerLM
    cfg = get_cfg(inner_size=256, tgt_length=16, mem_length=0, **over)
    with pytest.raises(CommuHipError) as e:
        MemTransformerLM(cfg, BaseVocab())
    assert frag in str(e.value) and "unsupported shape" in str(e.value)


def test_oracle_nucleus_filter_known_answers():
    """oracle.decode_ref.apply_top_p (the build's extra sampling mode): a token is kept while the mass before it, in order of
    decreasing probability with ties in id order, is < top_p."""
    import torch
    from oracle import decode_ref as Dz
    p = torch.tensor([0.05, 0.4, 0.1, 0.25, 0.2])
    assert Dz.apply_top_p(p, 1.0) is p
    assert torch.allclose(Dz.apply_top_p(p, 0.39), torch.tensor([0.0, 1.0, 0.0, 0.0, 0.0]))
    assert torch.allclose(Dz.apply_top_p(p, 0.5), torch.tensor([0.0, 0.4, 0.0, 0.25, 0.0]) / 0.65)
    assert torch.allclose(Dz.apply_top_p(p, 0.9), torch.tensor([0.0, 0.4, 0.1, 0.25, 0.2]) / 0.95)
    t = torch.tensor([0.2, 0.2, 0.2, 0.2, 0.2])                     # ties: lowest id first
    assert torch.allclose(Dz.apply_top_p(t, 0.5), torch.tensor([1.0, 1.0, 1.0, 0.0, 0.0]) / 3)


def test_bench_launch_mode_defaults():
    """bench.py's defaults: the step is replayed from hipGraphs only where it is host-bound (8 sequences per GPU, the tiny
    cfg-1 model); the headline, the merged default config and cfg-5 run eager (a replayed step is slower there, and a graph
    captured inside a timed region costs seconds).  The micro-batches of the released default config fold into one pass."""
    import argparse
    import bench
    base = dict(layers=6, d_model=512, heads=8, d_inner=1024, tgt_len=1024, mem_len=0, batch_per_gpu=64, batch_chunk=1,
                merge_chunks=None)
    ns = lambda **kw: argparse.Namespace(**{**base, **kw})
    assert not bench.auto_graph(ns())                                                        # the headline
    assert bench.auto_graph(ns(batch_per_gpu=8))                                             # strong scaling over 8 GPUs
    assert bench.auto_graph(ns(layers=2, d_model=128, heads=4, d_inner=256, tgt_len=256))    # cfg-1
    assert not bench.auto_graph(ns(layers=12, d_model=1024, heads=16, d_inner=2048, tgt_len=2048, mem_len=2048,
                                   batch_per_gpu=8))                                         # cfg-5
    default = ns(d_model=500, heads=10, d_inner=1000, tgt_len=128, mem_len=1024, batch_per_gpu=256, batch_chunk=4)
    assert bench.passes_of(default) == 1 and not bench.auto_graph(default)
    assert bench.passes_of(ns(batch_chunk=4, merge_chunks=False)) == 4
    assert bench.passes_of(ns(tgt_len=2048, batch_per_gpu=64, batch_chunk=2)) == 2          # 131 072 tokens: the loop


def test_bench_launches_its_own_ranks(monkeypatch):
    """`python bench.py --gpus N` with no torch.distributed environment starts ONE torch.distributed.run child for N ranks
    on 127.0.0.1 and returns its status; fewer visible GPUs than N is an error (never a one-rank number called N)."""
    import bench
    seen = {}

    class FakeProc:
        def __init__(self, cmd, env=None):
            seen["cmd"], seen["env"] = cmd, env

        def wait(self):
            return 7

    rc = bench.launch_ranks(4, ["--gpus", "4", "--steps", "3"], popen=FakeProc, device_count=8)
    assert rc == 7                                              # the child's status is the parent's
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    assert cmd[-4:] == ["--gpus", "4", "--steps", "3"] and cmd[-5].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" or "HSA_ENABLE_IPC_MODE_LEGACY" in os.environ
    seen.clear()
    assert bench.launch_ranks(2, ["--gpus", "2"], popen=FakeProc, device_count=1) != 0 and not seen      # too few GPUs: no launch


def test_profile_stamp_covers_every_kernel_source_and_the_traffic_switches(tmp_path, monkeypatch):
    """bench.py reports the HBM bytes of a committed PMC profile only when the profile's stamp matches the tree: the stamp
    hashes EVERY csrc/*.hip / *.h file plus ops.py and model.py (round 5's stamp skipped elementwise.hip, gemm.hip, ...), and
    the switches that change a step's traffic without changing a source file are part of the match."""
    import argparse
    import json
    import shutil
    import bench
    from commu_amd import source_stamp
    files = sorted(f for f in os.listdir(source_stamp.CSRC) if f.endswith((".hip", ".h")))
    assert {"elementwise.hip", "gemm.hip", "gemm8.hip", "relattn.hip", "relattn_kv3.hip", "band.hip", "common.h"} <= set(files)
    h0 = source_stamp.kernel_source_hash()
    # a change in ANY kernel source changes the stamp (here: a scratch copy of the source tree with one file touched)
    csrc2 = tmp_path / "csrc"
    shutil.copytree(source_stamp.CSRC, csrc2)
    monkeypatch.setattr(source_stamp, "CSRC", str(csrc2))
    assert source_stamp.kernel_source_hash() == h0
    with open(csrc2 / "elementwise.hip", "a") as f:
        f.write("\n// touched\n")
    assert source_stamp.kernel_source_hash() != h0
    monkeypatch.undo()
    # the committed round-6 profiles carry stamp + switches; a flipped switch refuses them
    args = argparse.Namespace(layers=6, d_model=512, heads=8, tgt_len=1024, mem_len=0, batch_per_gpu=64, batch_chunk=1,
                              merge_chunks=None)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "profiles", "r06_step_traffic.json")) as f:
        rec = json.load(f)
    assert set(rec["switches"]) == {"FWD_SAVES_P", "DELTA_KERNEL", "STORE_ATTN_P"}
    if rec["source_sha256"] == h0:          # (a later edit of a kernel source legitimately invalidates the profile)
        assert bench.step_traffic(args)["bytes"] == rec["step_hbm_bytes"]
        from commu_amd import ops
        monkeypatch.setattr(ops, "DELTA_KERNEL", not ops.DELTA_KERNEL)
        assert bench.step_traffic(args) is None and bench.pmc_traffic("commu_relattn_bwd_q", args) is None
    else:
        assert bench.step_traffic(args) is None


def test_strong_scaling_projection_arithmetic():
    """extra_rows.strong_b8...projection_8gpu_NOT_MEASURED: exchange = 2 (n-1)/n payload / (7 links x 153 GB/s x 0.7)."""
    import argparse
    import bench
    p = bench.strong_projection(argparse.Namespace(), ms_rank=4.0, tokens_rank=8192, n=8)
    want = 1e3 * 2 * 7 / 8 * 14.55e6 * 4 / (7 * 153e9 * 0.7)
    assert abs(p["fp32_serial"]["exchange_ms"] - want) < 2e-3 and abs(p["bf16_serial"]["exchange_ms"] - want / 2) < 2e-3
    assert p["fp32_overlapped"]["step_ms"] < p["fp32_serial"]["step_ms"] < 4.0 + want + 1e-3
    want_rate = 8 * 8192 / (p["fp32_serial"]["step_ms"] * 1e-3)          # (step_ms is rounded to a microsecond in the record)
    assert abs(p["fp32_serial"]["tokens_per_s_total"] - want_rate) < 2e-4 * want_rate
    assert "NOT" not in p["assumptions"] and "no straggler" in p["assumptions"]
