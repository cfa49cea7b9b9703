"""Batched autoregressive generation: `num_generate` sequences decoded in parallel on one GPU with a
per-layer K/V cache, ragged per-sequence lengths and the reference's chord-forcing rules applied
per sequence (commu/midi_generator/midi_inferrer.py:239-354; the reference itself generates one
sequence at a time, batch 1).

Per loop iteration every live sequence does exactly what one iteration of the reference's
`generate_sequence` does for it: at most one model step (with its memory kept or discarded,
quirks Q3/Q4) and at most one draw (from fresh logits, or from the logits it already divided by
the temperature when the previous draw was a rejected chord, quirk Q5).  The model steps of all
sequences that need one are ONE batched decode step; the draws are ONE sampling kernel; the token
ids come back with one host synchronisation per iteration.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import ops
from ._lib import call
from .midi_generator.midi_inferrer import TOKEN_OFFSET, TeacherForceTask

BF16, F32 = torch.bfloat16, torch.float32


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _s():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class DecodeState:
    """K/V caches + distance-indexed R tables for B sequences of up to Lmax positions."""

    def __init__(self, model, B: int, Lmax: int):
        fl = model._ensure_flat()
        dev = fl["dev"]
        self.model, self.B, self.Lmax = model, B, Lmax
        # kernel-side dimensions (zero-padded when the model's are not multiples of 64 / 32, see model.py)
        L, D = model.n_layer, model._Dp
        H, DH = model.n_head, model._DHp
        HD = H * DH
        self.kc = torch.zeros(L, B, H, Lmax, DH, device=dev, dtype=BF16)      # head-major: contiguous per (b, h)
        self.vc = torch.zeros(L, B, H, Lmax, DH, device=dev, dtype=BF16)
        self.klen = torch.zeros(B, device=dev, dtype=torch.int32)
        pd = ops.posemb(model.pos_emb.inv_freq, Lmax, model.d_model, ld=D)
        self.rd = [ops.gemm_nt(pd, model._weights(i)["r"]) for i in range(L)]
        self.logits = torch.zeros(B, 768, device=dev, dtype=F32)
        self.qkv = torch.zeros(B, 3 * HD, device=dev, dtype=BF16)
        self.vec = torch.zeros(B, HD, device=dev, dtype=BF16)

    def prefill(self, ctx: torch.Tensor):
        """ctx: int64 [T0, B] context tokens (midi_inferrer.py:186-197): fills the caches with their K/V
        (same kernels as training, memory-less forward) and sets klen = T0."""
        m = self.model
        T0, B = ctx.shape
        assert B == self.B
        _, _, qkvs = m._run_forward(ctx, None, None, None, need_grad=False, want_logits=True, want_kv=True)
        H, DH = m.n_head, m._DHp
        for i, qkv in enumerate(qkvs):
            kv = qkv.view(T0, B, 3, H, DH)
            self.kc[i, :, :, :T0].copy_(kv[:, :, 1].permute(1, 2, 0, 3))
            self.vc[i, :, :, :T0].copy_(kv[:, :, 2].permute(1, 2, 0, 3))
        self.klen.fill_(T0)

    # ---- hipGraph-captured step: the ~60 launches of one decode step + the sampling kernel replayed as one
    # graph from static input buffers (launch-bound otherwise: each kernel runs for only a few microseconds)
    def capture(self, temperature: float, top_k: int):
        dev = self.klen.device
        B = self.B
        self.g_tok = torch.zeros(B, dtype=torch.long, device=dev)
        self.g_active = torch.ones(B, dtype=torch.uint8, device=dev)
        self.g_keep = torch.ones(B, dtype=torch.uint8, device=dev)
        self.g_draw = torch.ones(B, dtype=torch.uint8, device=dev)
        self.g_uni = torch.full((B,), 0.5, device=dev)
        self.g_wrong = torch.zeros(B, 729, dtype=torch.uint8, device=dev)
        self.g_out = torch.zeros(B, dtype=torch.int32, device=dev)
        klen0 = self.klen.clone()

        def body():
            self.step(self.g_tok, self.g_active, self.g_keep)
            ops.sample_topk(self.logits, temperature, top_k, wrong=self.g_wrong, uniforms=self.g_uni,
                            active=self.g_draw, token=self.g_out)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            body()                                   # warm-up (allocations) outside capture
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            body()
        self.klen.copy_(klen0)                       # warm-up / capture advanced the lengths
        return self.graph

    def step(self, tokens: torch.Tensor, active: Optional[torch.Tensor], keep: torch.Tensor, want_logits=True):
        """One decode step for the sequences with active[b] != 0; klen advances where keep[b] != 0.
        tokens int64 [B]; active/keep uint8 [B].  Returns the fp32 logits buffer [B, 768] (rows of inactive
        sequences keep their previous content -- they may still be needed for a re-draw, quirk Q5)."""
        m = self.model
        dev = tokens.device
        B, L, H, DH, D = self.B, m.n_layer, m.n_head, m._DHp, m._Dp
        HD = H * DH
        h = ops.embed_fwd(tokens, m.word_emb.emb_layers[0].weight, ld=D)
        scale = m.attn_scale
        u, vb = m._uv()
        for i in range(L):
            w = m._weights(i)
            lay = m.layers[i]
            ops.gemm_nt(h, w["qkv"], out=self.qkv)
            call("commu_decode_attn", _p(self.qkv), self.qkv.stride(0), _p(self.kc[i]), _p(self.vc[i]),
                 _p(self.rd[i]), self.rd[i].stride(0), _p(u), _p(vb), _p(self.klen), _p(active),
                 _p(self.vec), self.vec.stride(0), B, H, DH, self.Lmax, scale, 1, _s())      # (K/V append fused in)
            z1 = ops.gemm_nt(self.vec, w["o"], resid=h)
            a, _, _ = ops.layernorm_fwd(z1, lay.dec_attn.layer_norm.weight, lay.dec_attn.layer_norm.bias)
            hid = ops.gemm_nt(a, w["w1"], bias=w["b1"], relu=True)
            z2 = ops.gemm_nt(hid, w["w2"], bias=w["b2"], resid=a)
            h, _, _ = ops.layernorm_fwd(z2, lay.pos_ff.layer_norm.weight, lay.pos_ff.layer_norm.bias)
        call("commu_decode_advance", _p(self.klen), _p(keep), B, self.Lmax, _s())
        if want_logits:
            V = m.n_token
            if active is None:
                ops.gemm_nt(h, m._emb_bf16(), out=self.logits[:, :V], bias=m.crit.out_layers[0].bias)
            else:       # only overwrite the rows of the sequences that stepped
                tmp = ops.gemm_nt(h, m._emb_bf16(), bias=m.crit.out_layers[0].bias,
                                  out=torch.empty(B, 768, device=dev, dtype=F32)[:, :V])
                self.logits[:, :V] = torch.where(active.bool()[:, None], tmp, self.logits[:, :V])
        return self.logits


class _Seq:
    __slots__ = ("seq", "teacher", "first", "done", "failed", "iters")


class BatchedGenerator:
    """Generates `len(input_datas)` sequences in parallel.  Each `input_data` needs temperature/top_k
    shared by the batch and per-sequence `num_measures` / `chord_token_components`."""

    def __init__(self, model, device, generation_length=4096, memory_length=4146):
        self.model, self.device = model, device
        self.generation_length, self.memory_length = generation_length, memory_length
        self.uniform_sources = None      # optional list of callables, one per sequence
        self.trace = None                # optional: per sequence list of (fed token, klen in, klen out)

    @torch.no_grad()
    def generate(self, encoded_metas: Sequence[Sequence[int]], input_datas, temperature: float, top_k: int):
        B = len(input_datas)
        dev = self.device
        state = DecodeState(self.model, B, min(self.memory_length + 1, 4224))
        n_cond = len(encoded_metas[0])
        ctx = torch.tensor([[0] + list(m[:n_cond - 1]) for m in encoded_metas], dtype=torch.long).t().contiguous()
        state.prefill(ctx.to(dev))
        seqs: List[_Seq] = []
        for b in range(B):
            s = _Seq()
            s.seq = [0] + list(encoded_metas[b][:n_cond])
            s.teacher = TeacherForceTask(input_datas[b])
            s.first, s.done, s.failed, s.iters = True, False, False, 0
            seqs.append(s)
        klen_host = [n_cond] * B
        rngs = self.uniform_sources or [np.random.RandomState(1000 + b).random_sample for b in range(B)]
        wrong = torch.zeros(B, TOKEN_OFFSET.VOCAB_SIZE, dtype=torch.uint8)
        for _ in range(self.generation_length):
            tok = torch.zeros(B, dtype=torch.long)
            active = torch.zeros(B, dtype=torch.uint8)
            keep = torch.zeros(B, dtype=torch.uint8)
            draw = torch.zeros(B, dtype=torch.uint8)
            uni = torch.full((B,), 0.5)
            any_live = False
            for b, s in enumerate(seqs):
                if s.done:
                    continue
                if s.iters >= self.generation_length or s.seq[-1] == 1:
                    s.done = True
                    continue
                any_live = True
                s.iters += 1
                t = s.teacher
                if t.next_tokens_forced:                              # midi_inferrer.py:247-251
                    s.seq.append(t.next_tokens_forced.pop(0))
                    tok[b], active[b], keep[b] = s.seq[-1], 1, 1
                    continue
                if t.no_sequence_appended:                            # :253-255 (re-draw, Q5)
                    t.no_sequence_appended = False
                elif s.first:                                         # :256-258 (memory discarded, Q3)
                    tok[b], active[b], keep[b] = s.seq[-1], 1, 0
                    s.first = False
                else:                                                 # :259-260 (Q4)
                    tok[b], active[b], keep[b] = s.seq[-1], 1, 1
                if not t.incomplete_filled:
                    t.incomplete_filled = s.seq.count(TOKEN_OFFSET.BAR) > 1
                if t.check_first_position(s.seq):
                    t.teach_first_position()
                    continue
                if t.check_one_chord_per_bar_case(s.seq) or t.check_mul_chord_per_bar_case(s.seq):
                    t.teach_chord_token()
                    continue
                draw[b] = 1
                if temperature != 0:
                    uni[b] = float(rngs[b]())
                wrong[b].zero_()
                if t.wrong_tokens:
                    wrong[b, list(t.wrong_tokens)] = 1
            if not any_live:
                break
            if self.trace is not None:
                for b in range(B):
                    if active[b]:
                        self.trace[b].append((int(tok[b]), klen_host[b], klen_host[b] + 1))
                        klen_host[b] += int(keep[b])
            if bool(active.any()):
                state.step(tok.to(dev), active.to(dev), keep.to(dev))
            if not bool(draw.any()):
                continue
            tokens = ops.sample_topk(state.logits, temperature, top_k, wrong=wrong.to(dev), uniforms=uni.to(dev),
                                     active=draw.to(dev)).cpu()          # the one sync of this iteration
            for b, s in enumerate(seqs):
                if not draw[b]:
                    continue
                token = int(tokens[b])
                t = s.teacher
                if token < 0:                                         # :286-291 sampling error -> sequence dropped
                    s.failed, s.done = True, True
                    continue
                if t.check_chord_position_passed(token):
                    t.teach_chord_position()
                elif t.check_wrong_chord_token_generated(token):
                    t.teach_wrong_chord_token(token)
                elif t.check_wrong_eos_generated(token):
                    t.teach_remnant_chord()
                elif t.check_wrong_bar_token_generated(token):
                    t.teach_eos()
                else:
                    s.seq.append(token)
        return [None if s.failed else s.seq for s in seqs], [s.teacher for s in seqs]
