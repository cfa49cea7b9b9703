#!/usr/bin/env python3
"""Benchmark of the ComMU Transformer-XL training hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W
N > 1 without a torch.distributed environment: this process starts the N ranks ITSELF (`python -m torch.distributed.run
--nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...` as a child, before touching the GPU), relays rank 0's line
and exits with the child's status; under `torch.distributed.run` (WORLD_SIZE set) it is one rank.  One rank per GPU over
RCCL; fewer visible GPUs than N is an error, never a silent one-rank number.  Prints ONE JSON line on rank 0.

Workload (BASELINE.json configs[1]): 6 layers, d_model 512, 8 heads (d_head 64), FFN 1024,
tgt_len 1024, mem_len 0, 64 sequences per GPU (weak scaling), bf16 GEMM/attention operands with
fp32 accumulation, fp32 master weights; one "step" = one optimiser step of the reference's
train() loop (forward + backward of every micro-batch, global-norm clip, Adam, LR schedule) on a
synthetic token batch already resident in HBM.  metric = non-pad target tokens per second over
all ranks (the reference's own tokens/s definition, train.py:157,174,186).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "commu-code_amd"))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

BF16_MFMA_PEAK_TFLOPS = 2500.0      # dense bf16, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0


def fwd_flops_per_token(L, D, DI, T, M, V=729):
    """F = L*[6D^2 + 4D^2*(M/T) + 2D^2 + 4*D*DI + 6*Kbar*D] + 2*D*V."""
    kbar = M + (T + 1) / 2.0
    return L * (6 * D * D + 4 * D * D * (M / T) + 2 * D * D + 4 * D * DI + 6 * kbar * D) + 2 * D * V


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(args):
    """The oracle's train step (oracle/xl_ref.py, a restatement of the reference pinned by the
    golden fixtures) timed on the host cores: same shape, micro-batch 2 (CPU tokens/s is roughly
    batch independent), 1 warm-up + timed steps bounded to ~20 s."""
    from oracle import xl_ref as X
    # 16 threads is the fastest setting for this step on the MI355X host (8: 1.7 s, 16: 1.4 s, 32: 1.4 s,
    # 64: 2.5 s, 256: 157 s per step -- measured with tests/probes/cpu_threads.py)
    nthreads = min(os.cpu_count(), 16)
    torch.set_num_threads(nthreads)
    s = X.XLShape(args.layers, args.heads, args.d_model, args.d_inner)
    p = X.init_params(s, 1)
    st = X.adam_init(p)
    Bc, T = 2, args.tgt_len
    g = torch.Generator().manual_seed(0)
    times = []
    t_start = time.time()
    for it in range(4):
        stream = torch.randint(2, 729, (T + 1, Bc), generator=g)
        t0 = time.time()
        X.train_step(p, st, s, stream[:-1], stream[1:], torch.zeros(Bc, dtype=torch.bool), [None], batch_chunk=1,
                     mem_len=args.mem_len, same_length=False, lr_now=1e-4, clip=1.0)
        dt = time.time() - t0
        if it > 0:
            times.append(dt)
        if time.time() - t_start > 25 and times:
            break
    per = sum(times) / len(times)
    return {"value": round(Bc * T / per, 1), "unit": "tokens/s", "cores": nthreads, "kind": "port",
            "cpu": f"{cpu_model()} ({os.cpu_count()} logical cpus visible)",
            "sample": f"{len(times)} optimiser steps of the same model shape at batch {Bc} x tgt_len {T} "
                      f"(fp32 PyTorch-CPU oracle, {per:.2f} s/step measured in this run on {nthreads} threads; the thread "
                      f"count comes from an earlier sweep, tests/probes/cpu_threads.py, not from this run)"}


def decode_bench(dev, args, klen0, steps=1024, B=64, graph=True, parity=False):
    """Second half of BASELINE.json's metric: autoregressive decode tokens/s.  The timed path is the one the
    generator ships (commu_amd.generate.ForcedDecoder): B sequences in parallel, ONE hipGraph replay per loop
    iteration = forcing decision kernel + K/V-cached decode step + temperature / top-k sampling kernel (top_k 32,
    T 0.95) + book-keeping kernel, the `done` flags polled every 16 iterations.  Random-init weights; the output bias
    keeps EOS / BAR / chord tokens from being drawn so that every iteration is one model step and one draw for all
    B sequences.  `roofline`: the HBM bytes one iteration must move (K and V caches once, the distance table and
    the weights once per step) / measured time, against the 8 TB/s peak."""
    import types
    from commu_amd.generate import ForcedDecoder
    from commu_amd.model.config_helper import get_cfg
    from commu_amd.model.dataset import BaseVocab
    from commu_amd.train import build_model
    cfg = get_cfg(num_layers=args.layers, num_heads=args.heads, units=args.d_model, inner_size=args.d_inner,
                  tgt_length=1, mem_length=4146, dropout=0.0, attention_dropout=0.0, same_length=True)
    model = build_model(cfg, BaseVocab(), dev, seed=1).eval()
    # parity: the fp32 mode of the generation path (model.parity_fp32: fp32 weights, activations, K/V cache and products --
    # the reference's arithmetic; not the headline, reported beside it)
    model.parity_fp32 = bool(parity)
    with torch.no_grad():
        bias = model.crit.out_layers[0].bias
        bias.zero_()
        bias[1:3] = -1e9
        bias[195:304] = -1e9
        dec = ForcedDecoder(model, B, generation_length=steps + 64, memory_length=4146, temperature=0.95, top_k=32)
        data = types.SimpleNamespace(num_measures=4.0, chord_token_components={"chord_token": [], "chord_position": []})
        meta = [574, 623, 627, 635, 639, 642, 651, 684, 694, 720, 727]          # README example
        uni = torch.rand(B, dec.ld_u).numpy()
        dec.load([meta] * B, [data] * B, uni)
        st = dec.state
        if klen0 > 11:
            # long memory: a REAL prefill of klen0 tokens (the meta context followed by random event tokens) through the
            # training kernels -- the K/V caches hold what generation would have written, not noise; the forcing state
            # records keep their own (sequence-length) counters, so only the cache length moves
            g = torch.Generator().manual_seed(3)
            ctx = torch.randint(3, 729, (klen0, B), generator=g)
            ctx[0] = 0
            ctx[1:11] = torch.tensor(meta[:10])[:, None]
            st.kc.zero_()
            st.vc.zero_()
            st.prefill(ctx.to(dev))
        if graph:
            dec.build_graph()

        dec.pre()

        def run(n):          # (ForcedDecoder.run's loop with a fixed iteration count: done flags polled one window late, no stall)
            done, live, pending = 0, True, False
            while done < n:
                # (what ForcedDecoder.run knows: the bound of the memory lengths and that all sequences are alive)
                dec.run_iterations(dec.POLL, graph, klen_bound=klen0 + 16 + done + dec.POLL, live_rows=B)
                done += dec.POLL
                if pending:
                    live = live and not bool(dec.poll_result()[:, 5].all())
                dec.poll_submit()
                pending = True
            live = live and not bool(dec.poll_result()[:, 5].all())
            return done, live
        run(16)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n, live = run(steps)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        assert live, "a bench sequence finished early"
    L, D, DI = args.layers, args.d_model, args.d_inner
    kmid = klen0 + 16 + n / 2.0
    nparam = L * (4 * D * D + 2 * D * DI) + 729 * D
    es = 4 if parity else 2          # bytes per cached / weight element
    bytes_step = B * L * 2 * kmid * D * es + L * kmid * D * es + nparam * es
    return {"tokens_per_s": round(B * n / dt, 1), "ms_per_step": round(1e3 * dt / n, 4), "sequences": B,
            "klen_start": klen0, "steps": n, "hipgraph": bool(graph), "dtype": "f32" if parity else "bf16",
            "roofline": {"bound": "hbm", "achieved": round(bytes_step / (dt / n) / 1e9, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(bytes_step / (dt / n) / 1e9 / HBM_PEAK_GBS, 4),
                         "bytes_per_step": int(bytes_step)}}


def decode_to_completion(dev, args, B=64):
    """the same 64-way generation run TO COMPLETION instead of a fixed number of iterations --
    ForcedDecoder.run() as generate.py uses it (graph replays, done flags polled every 16 iterations) until every sequence
    has drawn EOS (or a bar with no chords left, which forces EOS) or reached 4096 iterations.  Random-init weights: the
    lengths are geometric (a few hundred tokens), so the batch thins out as it goes -- tokens/s counts the tokens actually
    generated over the wall time of the whole call, including the iterations in which few sequences are still alive."""
    import types
    from commu_amd.generate import ForcedDecoder
    from commu_amd.model.config_helper import get_cfg
    from commu_amd.model.dataset import BaseVocab
    from commu_amd.train import build_model
    cfg = get_cfg(num_layers=args.layers, num_heads=args.heads, units=args.d_model, inner_size=args.d_inner,
                  tgt_length=1, mem_length=4146, dropout=0.0, attention_dropout=0.0, same_length=True)
    model = build_model(cfg, BaseVocab(), dev, seed=1).eval()
    with torch.no_grad():
        bias = model.crit.out_layers[0].bias
        bias.zero_()
        bias[195:304] = -1e9                      # no chord tokens (they would be rejected and redrawn: no progression given)
        dec = ForcedDecoder(model, B, generation_length=4096, memory_length=4146, temperature=0.95, top_k=32)
        data = types.SimpleNamespace(num_measures=4.0, chord_token_components={"chord_token": [], "chord_position": []})
        meta = [574, 623, 627, 635, 639, 642, 651, 684, 694, 720, 727]
        uni = torch.rand(B, dec.ld_u, generator=torch.Generator().manual_seed(5)).numpy()
        dec.load([meta] * B, [data] * B, uni)
        dec.run(use_graph=True)                   # builds the graph; a first, untimed generation
        dec.load([meta] * B, [data] * B, uni)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        dec.run(use_graph=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        seqs, _ = dec.sequences()
    lens = [len(s_) - 12 for s_ in seqs if s_ is not None]          # tokens generated after the 12-token context
    iters = int(dec.fsm[:, 7].max().item())
    out = {"tokens_per_s": round(sum(lens) / dt, 1), "wall_ms": round(1e3 * dt, 2), "sequences": B, "finished": len(lens),
           "generated_tokens": int(sum(lens)), "longest_sequence_iterations": iters,
           "mean_tokens_per_sequence": round(sum(lens) / max(1, len(lens)), 1), "hipgraph": True}
    # bulk request: 256 sequences through the same 64 slots.  Rounds of 64 (each round thins out towards its end) against
    # BatchedGenerator.generate_stream (a finished slot is re-armed with the next attempt at once: what generate.py does)
    from commu_amd.generate import BatchedGenerator
    gen = BatchedGenerator(model, dev, 4096, 4146)
    with torch.no_grad():
        ntok = [0]

        def accept(seq, rep):
            ntok[0] += 0 if seq is None else len(seq) - 12
            return True
        gen.generate_stream(meta, data, 0.95, 32, need=B, accept=accept, slots=B, seed=3)          # (graph build, untimed)
        ntok[0] = 0
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        got, started = gen.generate_stream(meta, data, 0.95, 32, need=4 * B, accept=accept, slots=B, seed=3)
        torch.cuda.synchronize()
        dts = time.perf_counter() - t0
        stream_tokens = sum(len(s_) - 12 for s_ in got)          # tokens of the 256 returned sequences only
        t0 = time.perf_counter()
        rtok = 0
        for r in range(4):
            gen.uniform_sources = [(lambda a=a: iter(BatchedGenerator.attempt_uniforms(3, a, 4097)).__next__)() for a in
                                   range(r * B, (r + 1) * B)]
            seqs_r, _ = gen.generate([meta] * B, [data] * B, 0.95, 32)
            rtok += sum(len(s_) - 12 for s_ in seqs_r if s_ is not None)
        torch.cuda.synchronize()
        dtr = time.perf_counter() - t0
    out["bulk_256_sequences"] = {
        "continuous_slots": {"tokens_per_s": round(stream_tokens / dts, 1), "wall_ms": round(1e3 * dts, 1),
                             "accepted": len(got), "attempts_started": started, "tokens_of_accepted": int(stream_tokens),
                             "tokens_of_all_attempts": int(ntok[0])},
        "rounds_of_64": {"tokens_per_s": round(rtok / dtr, 1), "wall_ms": round(1e3 * dtr, 1), "tokens": int(rtok)}}
    return out


def decode_cpu_baseline(args, steps=256):
    """The oracle's generation step (oracle/xl_ref.forward_generate: the reference's forward_generate restated, full
    QKV recomputation over the memory every step) + the oracle's sampling step, batch 1, sequential like the reference
    (midi_inferrer.py:199-237), timed on the host cores at memory length ~1000."""
    from oracle import decode_ref as Dz
    from oracle import xl_ref as X
    nthreads = min(os.cpu_count(), 16)
    torch.set_num_threads(nthreads)
    s = X.XLShape(args.layers, args.heads, args.d_model, args.d_inner)
    p = X.init_params(s, 1)
    g = torch.Generator().manual_seed(0)
    with torch.no_grad():
        ctx = torch.randint(2, 729, (1000, 1), generator=g)
        _, mems = X.forward_generate(p, s, ctx, None, 4146)
        tok = torch.randint(2, 729, (1, 1), generator=g)
        t0 = time.time()
        n = 0
        while n < steps and time.time() - t0 < 20:
            logits, mems = X.forward_generate(p, s, tok, mems, 4146)
            probs = Dz.apply_sampling(Dz.calc_probs(logits[-1, 0][1:].clone(), 0.95), 32, [])
            tok = torch.tensor([[Dz.draw_inverse_cdf(probs, 0.5)]])
            n += 1
        dt = time.time() - t0
    return {"value": round(n / dt, 1), "unit": "tokens/s", "cores": nthreads, "kind": "port",
            "sample": f"{n} sequential batch-1 steps at memory length 1000 (fp32 PyTorch-CPU oracle)"}


def kernel_source_hash():
    """Stamp of the code the committed PMC profiles must have been measured on (commu_amd/source_stamp.py: every kernel source and
    header, ops.py, model.py); the probes that write the profiles stamp the same value."""
    from commu_amd import source_stamp
    return source_stamp.kernel_source_hash()


def traffic_switches():
    from commu_amd import source_stamp
    return source_stamp.traffic_switches()


def pmc_traffic(kernel, args):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/rNN_pmc_traffic.json,
    produced by tests/probes/pmc_traffic.sh at the DEFAULT bench shape: FETCH_SIZE x 2 (gfx950 correction of
    MI355X_MICROARCH.md, HBM section) + WRITE_SIZE, both in KiB -> bytes).  None when no profile matches BOTH the
    shape and the hash of the current kernel sources (a stale number is refused, not reported)."""
    import glob
    shape = [args.layers, args.d_model, args.heads, args.tgt_len, args.mem_len, args.batch_per_gpu // passes_of(args)]
    try:
        sha, sw = kernel_source_hash(), traffic_switches()
    except OSError:
        return None
    if sw["FWD_SAVES_P"]:          # (the profiles are measured on the default path)
        return None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), reverse=True):
        try:
            with open(path) as f:
                rec = json.load(f)
        except (OSError, ValueError):
            continue
        if rec.get("shape") == shape and rec.get("source_sha256") == sha and rec.get("switches") == sw:
            return rec.get("bytes_per_launch", {}).get(kernel)
    return None


def step_traffic(args):
    """HBM bytes one optimiser step moves, from the committed whole-step PMC passes (profiles/rNN_step_traffic.json:
    tests/probes/step_traffic.sh -- every dispatch of the step, FETCH_SIZE x 2 + WRITE_SIZE, with a calibration line on a known
    1 GiB copy).  None unless a profile matches the shape AND the hash of the kernel sources it was measured on."""
    import glob
    shape = [args.layers, args.d_model, args.heads, args.tgt_len, args.mem_len, args.batch_per_gpu // passes_of(args)]
    try:
        sha, sw = kernel_source_hash(), traffic_switches()
    except OSError:
        return None
    if sw["FWD_SAVES_P"]:          # (the profiles are measured on the default path)
        return None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_step_traffic.json")), reverse=True):
        try:
            with open(path) as f:
                rec = json.load(f)
        except (OSError, ValueError):
            continue
        if rec.get("shape") == shape and rec.get("source_sha256") == sha and rec.get("switches") == sw:
            top = list(rec.get("kernels", {}).items())[:3]
            return {"bytes": rec.get("step_hbm_bytes"), "profile": os.path.basename(path),
                    "largest_movers": {k[:60]: round(v["bytes_per_step"] / 1e9, 3) for k, v in top}}
    return None


def passes_of(args):
    """Forward / backward passes per optimiser step: `batch_chunk`, or 1 when the Trainer folds the micro-batches into one
    pass (Trainer(merge_chunks=...): automatic up to Trainer.MERGE_MAX_ROWS tokens per step)."""
    from commu_amd.train import Trainer
    chunk = args.batch_chunk
    merge = getattr(args, "merge_chunks", None)
    if merge is None:
        merge = args.tgt_len * args.batch_per_gpu <= Trainer.MERGE_MAX_ROWS
    return 1 if (merge and 1 < chunk <= 16 and args.batch_per_gpu % chunk == 0) else chunk


def auto_graph(args):
    """hipGraph replay of the step pays when the step is host-bound: few tokens per pass through a small model (8 sequences
    x 1024 tokens per GPU at L6 D512: 4.7 instead of 6.1 ms).  With large kernels it does not (cfg-5: 61.5 / 71.9 ms
    replayed against 59.8 / 66.2 ms eager), so the default follows a work proxy: tokens per pass x layers x d_model^2
    (8 x 1024 x 6 x 512^2 = 1.3e10: graph; the headline 1.0e11, the merged default config 4.9e10, cfg-5 2.1e11: eager)."""
    return (args.batch_per_gpu // passes_of(args)) * args.tgt_len * args.layers * args.d_model ** 2 <= 2e10


LAST_COMM = {}          # rank-local GradReducer.comm_stats() of the last train_bench (N > 1 only)


def train_bench(args, dev, world, rank, steps, warmup, reset_prob=0.0):
    """W untimed + K timed optimiser steps of the shape in `args`; returns (elapsed seconds (max over ranks),
    tokens per step per rank, per-entry-point HIP-event times) -- None on ranks other than 0."""
    from commu_amd import _lib
    from commu_amd.ddp import GradReducer
    from commu_amd.model.config_helper import get_cfg
    from commu_amd.model.dataset import BaseVocab, synthetic_batch
    from commu_amd.train import Trainer, build_model

    B = args.batch_per_gpu
    cfg = get_cfg(num_layers=args.layers, num_heads=args.heads, units=args.d_model, inner_size=args.d_inner,
                  tgt_length=args.tgt_len, mem_length=args.mem_len, batch_size=B * world,
                  batch_chunk=args.batch_chunk, dropout=args.dropout, attention_dropout=args.dropout)
    model = build_model(cfg, BaseVocab(), dev, seed=cfg.TRAIN.seed)
    model.train()
    if args.no_side_stream:
        model.wgrad_side_stream = False
    if getattr(args, "fp8_forward", False):
        model.fp8_forward = True          # the layers' forward Linear products in MX-fp8 (BASELINE.json configs[4])
    reducer = GradReducer(wire_dtype=getattr(args, "grad_wire", "fp32")) if world > 1 else None
    if reducer is not None:
        reducer.broadcast_params(model)
    use_graph = getattr(args, "graph", None)
    if use_graph is None:
        use_graph = auto_graph(args)
    trainer = Trainer(model, cfg, num_gpus=world, reducer=reducer, graph=use_graph,
                      merge_chunks=getattr(args, "merge_chunks", None), settle_heap=True)
    batches = [synthetic_batch(args.tgt_len, B, dev, seed=cfg.TRAIN.seed + 1000 * rank + i, reset_prob=reset_prob)
               for i in range(4)]
    tokens_per_step = sum(b[3] for b in batches) // len(batches)

    step_log = [] if os.environ.get("BENCH_STEP_LOG") else None          # (diagnostics: host time of every step() call)
    gc_log = []
    if step_log is not None:          # ... and every collection of the Python garbage collector that falls into one
        import gc
        def _gc_cb(phase, info, _t=[0.0]):
            if phase == "start":
                _t[0] = time.perf_counter()
            else:
                gc_log.append((len(step_log), info["generation"], round(1e3 * (time.perf_counter() - _t[0]), 2)))
        gc.callbacks.append(_gc_cb)

    def run(nsteps, base, sample_events=False):
        for i in range(nsteps):
            d, t, r, n = batches[(base + i) % len(batches)]
            if step_log is not None:
                th = time.perf_counter()
            trainer.step(d, t, r, n)
            if step_log is not None:
                step_log.append(round(1e3 * (time.perf_counter() - th), 2))

    run(warmup, 0)
    if use_graph:
        # the capture + instantiation of the step's graphs (seconds) must not fall into the timed region: keep warming up
        # until the trainer replays (it captures once the XL memory is full and the optimiser state exists)
        extra = 0
        while trainer._graphs is None and trainer.graph_failed is None and extra < 8:
            run(1, warmup + extra)
            extra += 1
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(steps, warmup)          # the timed region: nothing but the K optimiser steps (no event pairs, no hooks)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if step_log is not None:
        ms = torch.cuda.memory_stats()
        print(f"[step log] {args.layers}x{args.d_model} T{args.tgt_len} M{args.mem_len} reset {reset_prob}: elapsed {elapsed:.3f} s; "
              f"host ms per step() {step_log}; gc (step, generation, ms) {[g for g in gc_log if g[2] >= 1.0]}; reserved {ms['reserved_bytes.all.current'] >> 20} MiB, segments "
              f"{ms['segment.all.allocated']}, retries {ms['num_alloc_retries']}", file=sys.stderr, flush=True)
        gc.callbacks.remove(_gc_cb)
    LAST_COMM.clear()
    if reducer is not None:
        # attribution for the scaling curve: wire bytes per rank and step, and the time the main stream sat in
        # GradReducer.finish() waiting for the last bucket (the part of the exchange the backward pass did not hide)
        LAST_COMM.update(reducer.comm_stats(last=steps))
    # Per-kernel HIP-event times come from SEPARATE steps right after the timed region (same trainer, same batches, same
    # streams): an event pair around each of the ~130 profiled calls costs ~4 % of a step, which used to sit inside the
    # number the driver reports (every fourth step).  hipGraph mode: a replay has no per-call hooks, so these steps run
    # eagerly -- same kernels, same streams.
    prof_names = ["commu_relattn_bwd_kv", "commu_relattn_bwd_q", "commu_relattn_fwd", "commu_relattn_fwd_save", "commu_gemm_nt_bf16",
                  "commu_gemm_tn_bf16", "commu_gemm_tn_bf16_grouped", "commu_relattn_bwd_band"]
    nprof = max(2, (steps + 3) // 4)
    _lib.profile_start(prof_names, shape_args={"commu_gemm_nt_bf16": (6, 7, 8)})          # (M, N, K) of every NT GEMM
    _lib.profile_enable(True)
    trainer.graph_mode = False
    run(nprof, warmup + steps)
    torch.cuda.synchronize()
    trainer.graph_mode = use_graph
    prof = _lib.profile_stop()
    if world > 1:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax)
    del trainer, model, batches
    torch.cuda.empty_cache()
    return elapsed, tokens_per_step, prof, steps / nprof


def attention_roofline(args, prof, tokens_per_step, elapsed, pscale=1.0):
    """Roofline of the dominant SINGLE kernel: the three attention entry points launch exactly one kernel at one
    shape (the GEMM entry points are a mix of shapes and tile variants; their share is in time_share).
    Algorithmic flops per launch = tokens * products * 2*Kbar*D  (Kbar = M + (T+1)/2):
      forward 3 products (QK^T, QR^T, PV); query-stationary backward 4 (QK^T, QR^T, dP, dQ);
      key-stationary backward 3 (dP, dV, dK: it re-reads the probabilities the query-stationary kernel stored; the
      d_head-32 variant, which recomputes QK^T and QR^T, is not on the bench path)."""
    from commu_amd import ops
    T, M = args.tgt_len, args.mem_len
    mb_tokens = tokens_per_step // passes_of(args)
    kbar = M + (T + 1) / 2.0
    # forward-saved probabilities (ops.FWD_SAVES_P, kernel-side d_head 64): the query-stationary backward kernel reads the
    # forward's probabilities instead of recomputing scores -- two products (dP, dq) instead of four, and a STREAMING kernel:
    # it is priced against the HBM roof below
    dh = args.d_model // args.heads
    fromp = bool(ops.FWD_SAVES_P and ops.STORE_ATTN_P and 32 < dh <= 64)
    products = {"commu_relattn_fwd": 3.0, "commu_relattn_bwd_q": 2.0 if fromp else 4.0, "commu_relattn_bwd_kv": 3.0}
    prof = dict(prof)          # (the training forward goes through the entry point that also saves its probabilities)
    prof["commu_relattn_fwd"] = list(prof.get("commu_relattn_fwd", [])) + list(prof.pop("commu_relattn_fwd_save", []))
    tot = {k: sum(v) for k, v in prof.items() if ":" not in k}
    cnt = {k: max(1, len(v)) for k, v in prof.items() if ":" not in k}
    dom = max(products, key=lambda k: tot.get(k, 0.0))
    fl_launch = mb_tokens * products[dom] * 2.0 * kbar * args.d_model
    avg_ms = tot[dom] / cnt[dom]
    achieved = fl_launch / (avg_ms * 1e-3) / 1e12
    share = {k: round(pscale * tot[k] / (1e3 * elapsed), 4) for k in tot if tot[k] > 0}
    if dom == "commu_relattn_bwd_q" and fromp and T % 64 == 0 and M % 64 == 0:
        # ALGORITHMIC bytes of one launch (every operand and result once): the saved probabilities
        # of the visited 64 x 64 tiles (4 tiles of 2176 bytes each), dO / K / V rows, the row statistics; written: P for the
        # key-stationary kernel and dS by distance (2 bytes per visited score each), dq
        Bm, H, HDk = mb_tokens // T, args.heads, args.heads * 64
        tiles = (T // 64) * (T // 64 + 1) // 2 + (T // 64) * (M // 64)          # visited (query tile, key tile) pairs per (batch, head)
        by = Bm * H * tiles * (4 * 2176 + 2 * 64 * 64 * 2) + (3 * T + 2 * M) * Bm * HDk * 2 + 2 * Bm * H * T * 4 + T * Bm * HDk * 2
        gbs = by / (avg_ms * 1e-3) / 1e9
        return {"kernel": dom, "bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": pmc_traffic(dom, args), "bytes_per_launch": int(by),
                "avg_launch_ms": round(avg_ms, 4), "launches": cnt[dom],
                "mfma": {"flops_per_launch": fl_launch, "achieved_tflops": round(achieved, 2),
                         "frac": round(achieved / BF16_MFMA_PEAK_TFLOPS, 4), "products": "dP = dO.V^T and dq = dS.K"},
                "why_hbm": "the forward pass saves its probabilities: this kernel recomputes no scores (2 of the former 4 "
                           "products) and streams ~2 GB per launch",
                "time_share": share, "event_sampling": "separate eager steps right after the timed region (max(2, K/4) of them)"}
    traffic = pmc_traffic(dom, args)
    out = {"kernel": dom, "bound": "mfma", "achieved": round(achieved, 2), "peak": BF16_MFMA_PEAK_TFLOPS,
           "unit": "TFLOP/s", "frac": round(achieved / BF16_MFMA_PEAK_TFLOPS, 4), "traffic": traffic,
           "flops_per_launch": fl_launch, "avg_launch_ms": round(avg_ms, 4), "launches": cnt[dom],
           "time_share": share, "event_sampling": "separate eager steps right after the timed region (max(2, K/4) of them)"}
    if traffic:          # the same launch against the OTHER roof: measured HBM bytes / this run's launch time
        gbs = traffic / (avg_ms * 1e-3) / 1e9
        out["hbm_view"] = {"measured_traffic_gbs": round(gbs, 1), "frac_of_hbm_peak": round(gbs / HBM_PEAK_GBS, 4)}
    return out


def iterator_bench(args, dev, steps, warmup, resident_ms):
    """The same optimiser steps fed by the batch producer instead of HBM-resident batches (the reference's tokens/s
    includes data loading, train.py:171-186): a synthetic ragged corpus is WRITTEN in the reference's on-disk format
    (input_/target_{train,val}.npy object arrays: 11 meta tokens, int16 events ending in EOS; preprocessor.py:161-162,
    dataset.py:74-87), read back by ComMUDataset and streamed through get_iterator (epoch scheduler, vectorised
    gather into pinned buffers on a worker thread, asynchronous H2D copies) into Trainer.step.  Reports the reference's
    metric (non-pad target tokens/s), the step time against the resident-batch step time of this run, and how long the
    training loop was blocked in next() (back-pressure from the producer's buffer ring; the GPU queue stays full)."""
    import shutil
    import tempfile

    import numpy as np
    from commu_amd.model.config_helper import get_cfg
    from commu_amd.model.dataset import BaseVocab, ComMUDataset
    from commu_amd.train import Trainer, build_model
    T, B = args.tgt_len, args.batch_per_gpu
    rng = np.random.default_rng(1111)
    tmp = tempfile.mkdtemp(prefix="commu_bench_npy_")
    try:
        for tag, nseq in (("train", max(600, 4 * B)), ("val", B + 8)):
            lens = rng.integers(T // 2, 3 * T, size=nseq)
            metas = np.empty(nseq, dtype=object)
            events = np.empty(nseq, dtype=object)
            for i, n in enumerate(lens):
                metas[i] = np.array(rng.integers(560, 729, size=11), dtype=object)
                ev = rng.integers(2, 560, size=int(n)).astype(np.int16)
                ev[-1] = 1
                events[i] = ev
            np.save(os.path.join(tmp, f"input_{tag}.npy"), metas, allow_pickle=True)
            np.save(os.path.join(tmp, f"target_{tag}.npy"), events, allow_pickle=True)
        cfg = get_cfg(num_layers=args.layers, num_heads=args.heads, units=args.d_model, inner_size=args.d_inner,
                      tgt_length=T, mem_length=args.mem_len, batch_size=B, batch_chunk=args.batch_chunk,
                      dropout=args.dropout, attention_dropout=args.dropout)
        ds = ComMUDataset(tmp, cfg)
        model = build_model(cfg, BaseVocab(), dev, seed=cfg.TRAIN.seed)
        model.train()
        trainer = Trainer(model, cfg, num_gpus=1, reducer=None)
        it = ds.get_iterator(B, T, dev, "train", True, seed=cfg.TRAIN.seed)()
        for _ in range(warmup):
            trainer.step(*next(it))
        torch.cuda.synchronize()
        wait, tokens = 0.0, 0
        t0 = time.perf_counter()
        for _ in range(steps):
            tw = time.perf_counter()
            d, t, r, n = next(it)
            wait += time.perf_counter() - tw
            trainer.step(d, t, r, n)
            tokens += n
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        it.close()
        del trainer, model
        torch.cuda.empty_cache()
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    ms = 1e3 * elapsed / steps
    return {"value": round(tokens / elapsed, 1), "unit": "non-pad target tokens/s", "ms_per_step": round(ms, 3),
            "slot_tokens_per_s": round(T * B * steps / elapsed, 1), "fill": round(tokens / (T * B * steps), 4),
            # time the loop was blocked in next(): BACK-PRESSURE, not starvation -- the producer's device buffers recycle at the
            # GPU's pace, so the host stays about one step ahead of the GPU instead of running several steps ahead
            "loop_wait_in_next_ms_per_step": round(1e3 * wait / steps, 4),
            "step_rate_vs_resident_batches": round(resident_ms / ms, 4), "steps": steps, "warmup": warmup,
            "corpus": "synthetic ragged (lengths uniform in [T/2, 3T)), on-disk .npy object arrays, shuffled epochs"}


def gemm_roofline(prof, elapsed, pscale=1.0):
    """One roofline row per NT GEMM shape of the step (commu_gemm_nt_bf16: every nn.Linear forward and every dX = dY . W;
    HIP events around each call on the launching stream, inside the timed region, side streams running):
    algorithmic flops 2 M N K / mean launch time against the dense bf16 MFMA peak.  Sorted by summed time."""
    rows = []
    for key, v in prof.items():
        if not key.startswith("commu_gemm_nt_bf16:") or not v:
            continue
        M, N, K = (int(x) for x in key.split(":")[1].split("x"))
        avg_ms = sum(v) / len(v)
        tf = 2.0 * M * N * K / (avg_ms * 1e-3) / 1e12
        rows.append({"shape_MNK": [M, N, K], "launches_sampled": len(v), "avg_launch_ms": round(avg_ms, 4),
                     "achieved": round(tf, 1), "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "bound": "mfma",
                     "frac": round(tf / BF16_MFMA_PEAK_TFLOPS, 4),
                     "time_share": round(pscale * sum(v) / (1e3 * elapsed), 4)})
    rows.sort(key=lambda r: -r["time_share"])
    return rows


# Further single-GPU rows (the shapes that were parity-tested but never timed); a few steps each
EXTRA_ROWS = [
    # tag, overrides
    # BASELINE.json configs[0] (the reference's own CPU-runnable case): the same shape on the GPU, with the oracle's CPU row
    ("cfg1_L2_D128_H4_DI256_T256_b64", dict(layers=2, d_model=128, heads=4, d_inner=256, tgt_len=256, mem_len=0,
                                           batch_per_gpu=64, cpu_row=True)),
    ("L6_D512_T1024_mem1024", dict(mem_len=1024)),
    ("reference_default_L6_D500_dh50_DI1000_T128_M1024_b256_chunk4",
     dict(d_model=500, heads=10, d_inner=1000, tgt_len=128, mem_len=1024, batch_per_gpu=256, batch_chunk=4)),
    ("cfg5_L12_D1024_H16_DI2048_T2048_M2048_b8_bf16",
     dict(layers=12, d_model=1024, heads=16, d_inner=2048, tgt_len=2048, mem_len=2048, batch_per_gpu=8)),
    # BASELINE.json configs[2] under the reference's STRONG-scaling semantics (train.py:396-397: batch_size // num_gpus columns per
    # rank): the per-rank work of the 8-GPU job -- 8 sequences x 1024 tokens, the step replayed from hipGraphs -- measured on ONE
    # GPU, with a PROJECTION (not a measurement: this pool has one GPU per box) of the 8-GPU step from it
    ("strong_b8_L6_D512_T1024_b8_per_gpu", dict(batch_per_gpu=8, strong_projection=True)),
    # (configs[4] names "fp8 MFMA GEMMs": the MX-fp8 forward path exists and is tested -- `--fp8-forward` -- but it is a
    #  measured LOSS at every shape of this model (63.5 vs 60.6 ms at this row in round 3): it is not a bench row)
]


def extra_rows(args, dev):
    rows = {}
    for tag, over in EXTRA_ROWS:
        a = argparse.Namespace(**{**vars(args), **over})
        # warm-up until the XL memory has reached its full length (tgt_len 128 / mem_len 1024: nine segments), so that
        # the timed steps run at the steady-state shapes
        steps, warmup = 8, max(8, a.mem_len // a.tgt_len + 6)          # (the caching allocator settles ~5 steps after the memory is full)
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        elapsed, tps, prof, pscale = train_bench(a, dev, 1, 0, steps, warmup)
        f = 3.0 * fwd_flops_per_token(a.layers, a.d_model, a.d_inner, a.tgt_len, a.mem_len) * tps
        rows[tag] = {"value": round(tps * steps / elapsed, 1), "unit": "tokens/s", "ms_per_step": round(1e3 * elapsed / steps, 3),
                     "steps": steps, "warmup": warmup, "tokens_per_step": tps, "batch_chunk": a.batch_chunk,
                     "passes_per_step": passes_of(a),
                     "step_mfma_frac": round(f / (elapsed / steps) / (BF16_MFMA_PEAK_TFLOPS * 1e12), 4),
                     "roofline": attention_roofline(a, prof, tps, elapsed, pscale)}
        if getattr(a, "cpu_row", False) and not args.no_cpu_baseline:
            rows[tag]["cpu_baseline"] = cpu_baseline(a)
        if getattr(a, "strong_projection", False):
            rows[tag]["launch"] = "hipGraph replay" if auto_graph(a) else "eager"
            rows[tag]["projection_8gpu_NOT_MEASURED"] = strong_projection(a, 1e3 * elapsed / steps, tps)
        if a.mem_len > 0 and not getattr(a, "fp8_forward", False):
            # reset_mems ~ Bernoulli(T / avg_len), avg_len = 512 (capped at 1), so that the reset
            # path (memory tiles skipped per column, zero-filled distances in the backward) is inside a timed region;
            # the row above, without resets, attends to the whole memory in every column (the heavier case)
            pr = min(1.0, a.tgt_len / 512.0)
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
            e2, tps2, _, _ = train_bench(a, dev, 1, 0, steps, warmup, reset_prob=pr)
            rows[tag]["with_resets"] = {"reset_prob": pr, "value": round(tps2 * steps / e2, 1),
                                        "ms_per_step": round(1e3 * e2 / steps, 3)}
        if a.batch_chunk > 1 and passes_of(a) == 1:
            # the same step as the reference's loop over micro-batches (Trainer(merge_chunks=False)), for comparison
            a2 = argparse.Namespace(**{**vars(a), "merge_chunks": False})
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
            e3, tps3, _, _ = train_bench(a2, dev, 1, 0, steps, warmup)
            rows[tag]["micro_batch_loop"] = {"passes_per_step": a.batch_chunk, "value": round(tps3 * steps / e3, 1),
                                             "ms_per_step": round(1e3 * e3 / steps, 3)}
    return rows


def strong_projection(a, ms_rank, tokens_rank, n=8):
    """PROJECTION (no multi-GPU hardware on this pool) of the n-GPU strong-scaling step of configs[2] from the measured
    per-rank step: the gradient exchange is one mean all-reduce of the flat fp32 gradient (14.55 M parameters = 58.2 MB; 29.1 MB
    on the bf16 wire), reduce-scatter + all-gather over the 7 xGMI links of a GPU (7 x 153 GB/s, MI355X_MICROARCH.md): each
    phase moves (n-1)/n of the payload per GPU.  `overlapped`: the exchange hides behind the backward pass except its last bucket
    (GradReducer: 16-MB buckets -> a quarter of the payload exposed); `serial`: nothing hidden; link efficiency 0.7 assumed."""
    params = 14.55e6
    link_gbs, links, eff = 153.0, 7, 0.7
    out = {"assumptions": f"{links} xGMI links x {link_gbs} GB/s per GPU at {eff} efficiency; ring-free full-mesh reduce-scatter + "
                          "all-gather; per-rank step as measured on one GPU (hipGraph replay); no straggler / launch skew term"}
    for wire, bytes_per in (("fp32", 4), ("bf16", 2)):
        payload = params * bytes_per
        t_ms = 1e3 * 2.0 * (n - 1) / n * payload / (links * link_gbs * 1e9 * eff)
        for mode, exposed in (("overlapped", 0.25), ("serial", 1.0)):
            step = ms_rank + exposed * t_ms
            out[f"{wire}_{mode}"] = {"exchange_ms": round(t_ms, 3), "step_ms": round(step, 3),
                                     "tokens_per_s_total": round(n * tokens_rank / (step * 1e-3), 1)}
    return out


def launch_ranks(n, argv, popen=None, device_count=None):
    """`bench.py --gpus n` outside torch.distributed.run: start the n ranks as ONE child (torch.distributed.run, which
    spawns them), wait, return its exit status.  Nothing here initialises the GPU (torch.cuda.device_count() does not on
    this image): the ranks are new processes, never an exec of a process that has touched the device.
    popen / device_count: test hooks."""
    import socket
    import subprocess
    ndev = torch.cuda.device_count() if device_count is None else device_count
    if ndev < n:
        print(f"bench.py: --gpus {n} but only {ndev} GPU(s) visible; not running a smaller job under that name",
              file=sys.stderr, flush=True)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    proc = (popen or subprocess.Popen)(cmd, env=env)
    return proc.wait()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--layers", type=int, default=6)
    ap.add_argument("--d-model", dest="d_model", type=int, default=512)
    ap.add_argument("--heads", type=int, default=8)
    ap.add_argument("--d-inner", dest="d_inner", type=int, default=1024)
    ap.add_argument("--tgt-len", dest="tgt_len", type=int, default=1024)
    ap.add_argument("--mem-len", dest="mem_len", type=int, default=0)
    ap.add_argument("--batch-per-gpu", type=int, default=64)
    ap.add_argument("--global-batch", type=int, default=0,
                    help="STRONG scaling: this many sequences over all ranks (the reference's semantics, train.py:396-397: "
                         "batch_size // num_gpus columns per rank); default 0 = weak scaling with --batch-per-gpu each")
    ap.add_argument("--batch-chunk", type=int, default=1)
    ap.add_argument("--dropout", type=float, default=0.1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-decode", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra single-GPU shape rows")
    ap.add_argument("--fp8-forward", dest="fp8_forward", action="store_true",
                    help="forward Linear products of the layers in MX-fp8 (opt-in; bf16 is the default and the headline)")
    ap.add_argument("--graph", dest="graph", action=argparse.BooleanOptionalAction, default=None,
                    help="replay the optimiser step from hipGraphs (Trainer(graph=True)); --no-graph: eager launches; "
                         "default: graphs when a pass is small (tokens x layers x d_model^2 <= 2e10: there the eager step is bound "
                         "by the host's launch rate; with large kernels the eager step with its side streams is faster)")
    ap.add_argument("--from-iterator", dest="from_iterator", action="store_true",
                    help="also time the step fed by ComMUDataset.get_iterator from an on-disk .npy corpus (on by default "
                         "with the extra rows)")
    ap.add_argument("--merge-chunks", dest="merge_chunks", action=argparse.BooleanOptionalAction, default=None,
                    help="run the batch_chunk micro-batches of a step as one pass with per-micro-batch loss weights "
                         "(default: automatic, up to 65536 tokens per step); --no-merge-chunks: the reference's loop")
    ap.add_argument("--grad-wire", dest="grad_wire", choices=("fp32", "bf16"), default="fp32",
                    help="gradient exchange format of an N > 1 job: fp32 mean all-reduce (default, DDP's arithmetic) or bf16 "
                         "on the wire with fp32 summation (all-to-all + all-gather, half the bytes)")
    ap.add_argument("--no-side-stream", action="store_true",
                    help="weight-gradient work on the main stream (default: a side stream)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", init_method="env://")
        world = dist.get_world_size()                     # n_gpus of the line is what really runs

    scaling = "weak"
    if args.global_batch > 0:
        if args.global_batch % world:
            raise SystemExit(f"--global-batch {args.global_batch} is not divisible by {world} ranks (train.py:396)")
        args.batch_per_gpu, scaling = args.global_batch // world, "strong"
    B = args.batch_per_gpu
    elapsed, tokens_per_step, prof, pscale = train_bench(args, dev, world, rank, args.steps, args.warmup)
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    ms_per_step = 1e3 * elapsed / args.steps
    value = tokens_per_step * world * args.steps / elapsed
    L, D, DI, T, M, H = args.layers, args.d_model, args.d_inner, args.tgt_len, args.mem_len, args.heads
    f_fwd = fwd_flops_per_token(L, D, DI, T, M)
    step_flops = 3.0 * f_fwd * tokens_per_step
    out = {
        "metric": "training tokens/sec at d_model=512 tgt_len=1024",
        "value": round(value, 1), "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
        "dtype": "bf16", "data": "synthetic",
        "config": {"workload": f"transformer-xl train step L{L} D{D} H{H} DI{DI} tgt_len{T} mem_len{M} vocab729",
                   "global_batch": B * world, "batch_per_gpu": B, "batch_chunk": args.batch_chunk,
                   "passes_per_step": passes_of(args), "seq_len": T,
                   "dropout": args.dropout, "parallelism": f"dp{world}", "optimizer": "clip1.0+Adam+invsqrt-LR",
                   "launch": ("hipGraph replay (the per-kernel events come from eager steps after the timed region)"
                              if (args.graph if args.graph is not None else auto_graph(args)) else "eager"),
                   "weights": "random init (train.py:291-342)"},
        "step_tflops_algorithmic": round(step_flops / 1e12, 3),
        "step_mfma_frac": round(step_flops / (elapsed / args.steps) / (BF16_MFMA_PEAK_TFLOPS * 1e12), 4),
        "roofline": attention_roofline(args, prof, tokens_per_step, elapsed, pscale),
        "roofline_gemm": gemm_roofline(prof, elapsed, pscale),
    }
    # the step's SECOND roofline: measured HBM bytes per step (PMC, committed profile) / this run's step time against the
    # 8 TB/s peak -- the K = 512 GEMMs, LayerNorms and the attention scratch are memory-side work the MFMA fraction hides
    st = step_traffic(args) if world == 1 else None
    out["step_hbm_bytes"] = None if st is None else st["bytes"]
    out["step_hbm_frac"] = (None if st is None or not st["bytes"] else
                            round(st["bytes"] / (elapsed / args.steps) / (HBM_PEAK_GBS * 1e9), 4))
    if st is not None:
        out["step_hbm_profile"] = {"file": st["profile"], "largest_movers_GB_per_step": st["largest_movers"]}
    if world > 1:
        # what the exchange cost THIS rank (rank 0): bytes it put on the links per step and the un-hidden wait
        c = dict(LAST_COMM)
        c["exposed_share_of_step"] = (round(c["exposed_ms_per_step"] / ms_per_step, 4)
                                      if c.get("exposed_ms_per_step") is not None else None)
        c["payload_bytes"] = None if not c else int(c["wire_bytes_per_step"] * world / (2 * (world - 1)))
        out["comm"] = c
    if world == 1 and (args.from_iterator or not args.no_extra):
        out["iterator_fed"] = iterator_bench(args, dev, args.steps, args.warmup, ms_per_step)
    if world == 1 and not args.no_extra:
        out["extra_rows"] = extra_rows(args, dev)
    if world == 1 and not args.no_decode:
        out["decode"] = {"metric": "autoregressive decode tokens/sec (64 sequences in parallel, device-resident forcing "
                                   "+ K/V-cache step + top-k 32 / T 0.95 sampling in one hipGraph per iteration)",
                         "short_memory": decode_bench(dev, args, 11), "long_memory": decode_bench(dev, args, 1000),
                         "long_memory_no_graph": decode_bench(dev, args, 1000, steps=128, graph=False),
                         "to_completion": decode_to_completion(dev, args),
                         # the fp32 parity mode (greedy tokens equal to the reference's wherever the top-1 / top-2 gap exceeds fp32 summation-order noise): same loop, fp32 operands
                         "short_memory_parity_fp32": decode_bench(dev, args, 11, steps=256, parity=True)}
        if not args.no_cpu_baseline:
            out["decode"]["cpu_baseline"] = decode_cpu_baseline(args)
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args)
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
