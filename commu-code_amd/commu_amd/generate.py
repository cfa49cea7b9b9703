"""Batched autoregressive generation: `num_generate` sequences decoded in parallel on one GPU with a per-layer
K/V cache, ragged per-sequence lengths and the reference's chord / bar forcing applied per sequence
(commu/midi_generator/midi_inferrer.py:239-354; the reference itself generates one sequence at a time, batch 1).

The whole loop iteration lives on the device: the forcing rules are a per-sequence state record advanced by two
small kernels (csrc/forcing.hip), between them the batched decode step and the sampling kernel.  One iteration =
one replay of a hipGraph; the host only polls the `done` flags every few iterations.  Per iteration every live
sequence does exactly what one iteration of the reference's `generate_sequence` does for it: at most one model
step (memory kept or discarded, quirks Q3/Q4) and at most one draw (from fresh logits, or from the logits already
divided by the temperature when the previous draw was a rejected chord, quirk Q5).
"""
from __future__ import annotations

import ctypes as C
import math
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import ops
from ._lib import CommuHipError, call
from .midi_generator.midi_inferrer import TOKEN_OFFSET, ForcingReport

BF16, F32 = torch.bfloat16, torch.float32
VPAD = 768
# decode step: everything after a layer's attention as one launch (commu_decode_layer_tail) where the shape is supported;
# False: one launch per Linear / LayerNorm (the only path for other shapes)
USE_LAYER_TAIL = True


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _s():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class DecodeState:
    """K/V caches + distance-indexed R tables for B sequences of up to Lmax positions."""

    MAX_POSITIONS = 4224          # cache rows the decode attention kernel supports (commu_decode_attn)

    def __init__(self, model, B: int, Lmax: int):
        if Lmax > self.MAX_POSITIONS:
            raise CommuHipError(f"decode cache of {Lmax} positions requested; this build supports {self.MAX_POSITIONS} "
                                "(the reference's 1 + 4146 fits)")
        if getattr(model, "ffn_activation", "relu") != "relu":
            raise CommuHipError("the cached decode step is built for the reference's ReLU FFN only")
        self.model, self.B, self.Lmax = model, B, Lmax
        self.attn_splits, self.split_ws, self.split_cnt = 1, None, None
        # fp32 parity mode (model.parity_fp32, read when the state is built): fp32 weights, activations and K/V cache, the
        # kernels of csrc/parity_f32.hip -- the mode that carries the "bit-exact greedy tokens" claim
        self.parity = bool(getattr(model, "parity_fp32", False))
        if self.parity:
            return self._init_f32()
        fl = model._ensure_flat()
        dev = fl["dev"]
        # kernel-side dimensions (zero-padded when the model's are not multiples of 64 / 32, see model.py)
        L, D = model.n_layer, model._Dp
        H, DH = model.n_head, model._DHp
        HD = H * DH
        self.kc = torch.zeros(L, B, H, Lmax, DH, device=dev, dtype=BF16)      # head-major: contiguous per (b, h)
        self.vc = torch.zeros(L, B, H, Lmax, DH, device=dev, dtype=BF16)
        self.klen = torch.zeros(B, device=dev, dtype=torch.int32)
        self._pd = ops.posemb(model.pos_emb.inv_freq, Lmax, model.d_model, ld=D, clamp_len=int(model.clamp_len))
        self.rd = [ops.gemm_nt(self._pd, model._weights(i)["r"]) for i in range(L)]
        self.logits = torch.zeros(B, VPAD, device=dev, dtype=F32)          # persistent: re-draws read them again (Q5)
        self.logits_new = torch.zeros(B, VPAD, device=dev, dtype=F32)      # this step's logits before the row select
        self.qkv = torch.zeros(B, 3 * HD, device=dev, dtype=BF16)
        self.vec = torch.zeros(B, HD, device=dev, dtype=BF16)
        # layer-tail launches: hand-off buffers per layer, arrival counters per launch, give-up flag
        DI = model._DIp
        self.tail_ok = bool(call("commu_decode_tail_supported", B, D, DI, HD)) and L > 0
        if self.tail_ok:
            nw = call("commu_decode_tail_sync_words")
            self.t_z1 = torch.zeros(L, B, D, device=dev, dtype=BF16)
            self.t_hid = torch.zeros(L, B, DI, device=dev, dtype=BF16)
            self.t_z2 = torch.zeros(L, B, D, device=dev, dtype=BF16)
            self.t_h = torch.zeros(L, B, D, device=dev, dtype=BF16)
            self.t_h0 = torch.zeros(B, D, device=dev, dtype=BF16)
            self.t_sync = torch.zeros(L, nw, device=dev, dtype=torch.int32)
            self.t_err = torch.zeros(1, device=dev, dtype=torch.int32)
            self.t_packs = None
            self.repack()

    # ---- fp32 parity mode ------------------------------------------------------------------------------------------
    def _init_f32(self):
        m = self.model
        dev = next(m.parameters()).device
        if dev.type != "cuda":
            raise CommuHipError("the decode state lives on an MI355X (no CPU fallback)")
        B, Lmax = self.B, self.Lmax
        L, D, DI, H, DH, V = m.n_layer, m.d_model, m.d_inner, m.n_head, m.d_head, m.n_token
        HD = H * DH
        z = lambda *shape: torch.zeros(*shape, device=dev, dtype=F32)
        self.kc, self.vc = z(L, B, Lmax, HD), z(L, B, Lmax, HD)          # [layer][sequence][position][head * d_head]
        self.klen = torch.zeros(B, device=dev, dtype=torch.int32)
        self._pd = ops.posemb_f32(m.pos_emb.inv_freq, Lmax, D, clamp_len=int(m.clamp_len))
        self.rd = [z(Lmax, HD) for _ in range(L)]
        self.logits, self.logits_new = z(B, VPAD), z(B, VPAD)
        self.tail_ok = False
        self.f32 = {"h0": z(B, D), "qkv": z(B, 3 * HD), "vec": z(B, HD), "z1": z(B, D), "a": z(B, D), "hid": z(B, DI),
                    "z2": z(B, D), "h": [z(B, D) for _ in range(L)]}
        self.repack()

    def _prefill_f32(self, ctx):
        m = self.model
        T0, B = ctx.shape
        _, _, qkvs = m._run_forward_f32(ctx, None, want_kv=True)
        HD = m.n_head * m.d_head
        for i, qkv in enumerate(qkvs):
            kv = qkv.view(T0, B, 3, HD)
            self.kc[i, :, :T0].copy_(kv[:, :, 1].permute(1, 0, 2))
            self.vc[i, :, :T0].copy_(kv[:, :, 2].permute(1, 0, 2))
        self.klen.fill_(T0)

    def _step_f32(self, tokens, active, keep, want_logits):
        """step() on fp32 operands: per layer [qkv_net, K/V append, cached attention over the ragged memories, o_net +
        residual, LayerNorm, FFN, LayerNorm], then the tied output layer (model.py:283-352,163-181,46).  Static buffers
        only: capturable."""
        m, f = self.model, self.f32
        B, L, H, DH = self.B, m.n_layer, m.n_head, m.d_head
        HD, V = H * DH, m.n_token
        E = m.word_emb.emb_layers[0].weight
        h = ops.embed_f32(tokens, E, out=f["h0"])
        u, vb = m.r_w_bias, m.r_r_bias
        for i in range(L):
            lay = m.layers[i]
            att, ff = lay.dec_attn, lay.pos_ff
            ops.gemm_nt_f32(h, att.qkv_net.weight, out=f["qkv"])
            ops.decode_kv_append_f32(f["qkv"], self.kc[i], self.vc[i], self.klen, active, HD, self.Lmax)
            ops.relattn_f32(f["qkv"][:, :HD], self.kc[i], self.vc[i], HD, self.Lmax * HD, self.rd[i], u, vb, 1, 0, B, H, DH,
                            bool(m.same_length), int(m.mem_len), m.attn_scale, klen=self.klen, out=f["vec"])
            ops.gemm_nt_f32(f["vec"], att.o_net.weight, resid=h, out=f["z1"])
            ops.layernorm_f32(f["z1"], att.layer_norm.weight, att.layer_norm.bias, att.layer_norm.eps, out=f["a"])
            ops.gemm_nt_f32(f["a"], ff.CoreNet[0].weight, bias=ff.CoreNet[0].bias, relu=True, out=f["hid"])
            ops.gemm_nt_f32(f["hid"], ff.CoreNet[3].weight, bias=ff.CoreNet[3].bias, resid=f["a"], out=f["z2"])
            h = ops.layernorm_f32(f["z2"], ff.layer_norm.weight, ff.layer_norm.bias, ff.layer_norm.eps, out=f["h"][i])
        if keep is not None:
            call("commu_decode_advance", _p(self.klen), _p(keep), B, self.Lmax, _s())
        if want_logits:
            dst = self.logits if active is None else self.logits_new
            ops.gemm_nt_f32(h, E, bias=m.crit.out_layers[0].bias, out=dst[:, :V])
            if active is not None:
                call("commu_copy_rows_masked_f32", _p(self.logits), VPAD, _p(self.logits_new), VPAD, _p(active), B, V, _s())
        return self.logits

    def prefill(self, ctx: torch.Tensor):
        """ctx: int64 [T0, B] context tokens (midi_inferrer.py:186-197): fills the caches with their K/V
        (same kernels as training, memory-less forward) and sets klen = T0."""
        m = self.model
        T0, B = ctx.shape
        assert B == self.B
        if T0 >= self.Lmax:
            raise CommuHipError(f"context of {T0} tokens does not fit a decode cache of {self.Lmax} positions")
        if self.parity:
            return self._prefill_f32(ctx)
        _, _, qkvs = m._run_forward(ctx, None, None, None, need_grad=False, want_logits=True, want_kv=True)
        H, DH = m.n_head, m._DHp
        for i, qkv in enumerate(qkvs):
            kv = qkv.view(T0, B, 3, H, DH)
            self.kc[i, :, :, :T0].copy_(kv[:, :, 1].permute(1, 2, 0, 3))
            self.vc[i, :, :, :T0].copy_(kv[:, :, 2].permute(1, 2, 0, 3))
        self.klen.fill_(T0)

    def _attn(self, i, u, vb, active, B, H, DH, scale):
        """Cached attention of layer i (K/V append fused in).  attn_splits > 1: the keys of a (sequence, head) pair over
        several workgroups (long memories, few live sequences); pairs with fewer than 512 keys run unsplit either way."""
        if self.attn_splits > 1:
            if self.split_ws is None:
                self.split_ws = torch.empty(B * H * 16 * (DH + 2), device=self.qkv.device, dtype=torch.float32)
                self.split_cnt = torch.zeros(B * H, device=self.qkv.device, dtype=torch.int32)
            call("commu_decode_attn_split", _p(self.qkv), self.qkv.stride(0), _p(self.kc[i]), _p(self.vc[i]),
                 _p(self.rd[i]), self.rd[i].stride(0), _p(u), _p(vb), _p(self.klen), _p(active),
                 _p(self.vec), self.vec.stride(0), B, H, DH, self.Lmax, scale, 1, int(self.attn_splits),
                 _p(self.split_ws), _p(self.split_cnt), _s())
        else:
            call("commu_decode_attn", _p(self.qkv), self.qkv.stride(0), _p(self.kc[i]), _p(self.vc[i]),
                 _p(self.rd[i]), self.rd[i].stride(0), _p(u), _p(vb), _p(self.klen), _p(active),
                 _p(self.vec), self.vec.stride(0), B, H, DH, self.Lmax, scale, 1, _s())

    def step(self, tokens: torch.Tensor, active: Optional[torch.Tensor], keep: Optional[torch.Tensor], want_logits=True):
        """One decode step for the sequences with active[b] != 0; klen advances where keep[b] != 0 (keep = None: the
        caller advances the lengths itself, ForcedDecoder does it in its book-keeping kernel).
        tokens int64 [B]; active/keep uint8 [B].  Returns the fp32 logits buffer [B, 768] (rows of inactive
        sequences keep their previous content -- they may still be needed for a re-draw, quirk Q5).
        No host synchronisation, no data-dependent allocation: safe inside a hipGraph capture.

        Kernel chain per layer: QKV Linear -> cached attention (K/V append fused in) -> o_net Linear + residual ->
        FFN Linear 1 -> FFN Linear 2 + residual; the two LayerNorms (model.py:352,179) run INSIDE the Linear that
        consumes them (commu_gemm_nt_ln_bf16), which also stores the normalised rows for the next residual add."""
        m = self.model
        if self.parity:
            return self._step_f32(tokens, active, keep, want_logits)
        B, L, H, DH, D = self.B, m.n_layer, m.n_head, m._DHp, m._Dp
        if USE_LAYER_TAIL and self.tail_ok:
            return self._step_tail(tokens, active, keep, want_logits)
        h = ops.embed_fwd(tokens, m.word_emb.emb_layers[0].weight, ld=D)
        scale = m.attn_scale
        u, vb = m._uv()
        z2, ln2 = None, None          # pre-LayerNorm output of the previous layer's FFN and that LayerNorm
        for i in range(L):
            w = m._weights(i)
            lay = m.layers[i]
            if z2 is None:
                ops.gemm_nt(h, w["qkv"], out=self.qkv)
            else:                     # h = LN2(z2) of the layer below, computed on the fly and stored
                h = torch.empty_like(z2)
                ops.gemm_nt_ln(z2, ln2.weight, ln2.bias, w["qkv"], out=self.qkv, a_out=h, eps=ln2.eps)
            self._attn(i, u, vb, active, B, H, DH, scale)                                    # (K/V append fused in)
            z1 = ops.gemm_nt(self.vec, w["o"], resid=h)
            ln1 = lay.dec_attn.layer_norm
            a = torch.empty_like(z1)
            hid = ops.gemm_nt_ln(z1, ln1.weight, ln1.bias, w["w1"], a_out=a, bias=w["b1"], relu=True, eps=ln1.eps)
            z2 = ops.gemm_nt(hid, w["w2"], bias=w["b2"], resid=a)
            ln2 = lay.pos_ff.layer_norm
        if keep is not None:
            call("commu_decode_advance", _p(self.klen), _p(keep), B, self.Lmax, _s())
        if want_logits:
            V = m.n_token
            dst = self.logits if active is None else self.logits_new
            ops.gemm_nt_ln(z2, ln2.weight, ln2.bias, m._emb_bf16(), out=dst[:, :V], bias=m.crit.out_layers[0].bias,
                           eps=ln2.eps)
            if active is not None:       # only the rows of the sequences that stepped are replaced
                call("commu_copy_rows_masked_f32", _p(self.logits), VPAD, _p(self.logits_new), VPAD, _p(active),
                     B, V, _s())
        return self.logits


    def repack(self):
        """Refresh what the decode step derives from the weights -- the distance tables and the packed weight copies the
        layer-tail launches read (commu_decode_tail_pack): call again whenever the model's weights changed.  Everything is
        rewritten IN PLACE: a captured graph keeps pointing at the same buffers."""
        m = self.model
        if self.parity:
            for i in range(m.n_layer):
                ops.gemm_nt_f32(self._pd, m.layers[i].dec_attn.r_net.weight, out=self.rd[i])
            return
        for i in range(m.n_layer):          # the distance tables r_net(pos_emb) depend on the weights too
            ops.gemm_nt(self._pd, m._weights(i)["r"], out=self.rd[i])
        if not self.tail_ok:
            return

        def pack(wt, out):
            n, k = wt.shape
            if out is None:
                out = torch.empty(call("commu_decode_tail_pack_bytes", n, k) // 2, device=wt.device, dtype=BF16)
            call("commu_decode_tail_pack", _p(wt), wt.stride(0), n, k, _p(out), _s())
            return out
        old = self.t_packs
        packs = []
        for i in range(m.n_layer):
            w = m._weights(i)
            packs.append({k: pack(w[k], None if old is None else old[i][k]) for k in ("qkv", "o", "w1", "w2")})
        self.t_packs = packs
        self.t_pack_e = pack(m._emb_bf16(), None if old is None else self.t_pack_e)

    def _step_tail(self, tokens, active, keep, want_logits):
        """step() with one launch per layer after the attention (csrc/decode_tail.hip): embedding, layer 0's QKV Linear,
        then per layer [cached attention, layer tail]; the tail of layer i ends with layer i + 1's QKV Linear, the last
        one with the logits."""
        m = self.model
        B, L, H, DH, D = self.B, m.n_layer, m.n_head, m._DHp, m._Dp
        HD, DI = H * DH, m._DIp
        V = m.n_token
        E = m.word_emb.emb_layers[0].weight
        scale = m.attn_scale
        u, vb = m._uv()
        ws = [m._weights(i) for i in range(L)]
        h = self.t_h0
        pk = self.t_packs
        call("commu_decode_head", _p(tokens), _p(E), E.shape[1], E.shape[0], math.sqrt(E.shape[1]), _p(pk[0]["qkv"]),
             _p(h), D, _p(self.qkv), self.qkv.stride(0), B, D, DI, HD, _p(self.t_sync), self.t_sync.numel(), _s())
        dst = self.logits          # (the logits launch skips the rows of the sequences that did not step)
        for i in range(L):
            w, lay = ws[i], m.layers[i]
            self._attn(i, u, vb, active, B, H, DH, scale)
            last = i == L - 1
            if last and not want_logits:
                break
            ln1, ln2 = lay.dec_attn.layer_norm, lay.pos_ff.layer_norm
            if last:
                wn, nn_, bn, out_n, ld_on, h_out = self.t_pack_e, V, m.crit.out_layers[0].bias, dst, dst.stride(0), None
            else:
                wn, nn_, bn, out_n, ld_on, h_out = pk[i + 1]["qkv"], 3 * HD, None, self.qkv, self.qkv.stride(0), self.t_h[i]
            call("commu_decode_layer_tail", _p(self.vec), self.vec.stride(0), _p(h), h.stride(0),
                 _p(pk[i]["o"]), _p(pk[i]["w1"]), _p(w["b1"]), _p(pk[i]["w2"]), _p(w["b2"]), _p(ln1.weight), _p(ln1.bias),
                 float(ln1.eps), _p(ln2.weight), _p(ln2.bias), float(ln2.eps), ln1.weight.numel(), _p(wn), nn_, _p(bn),
                 1 if last else 0, _p(active), _p(self.t_z1[i]), _p(self.t_hid[i]), _p(self.t_z2[i]), _p(h_out), D,
                 _p(out_n), ld_on, B, D, DI, HD, _p(self.t_sync[i]), _p(self.t_err), _s())
            h = h_out
        if keep is not None:
            call("commu_decode_advance", _p(self.klen), _p(keep), B, self.Lmax, _s())
        return self.logits

    def check(self):
        """Raises when a layer-tail launch gave up waiting for its peers (one D2H read: call it outside the loop)."""
        if self.tail_ok and int(self.t_err.item()) != 0:
            raise CommuHipError(f"decode layer-tail launch timed out at hand-off {int(self.t_err.item())}: its results are "
                                "invalid")


class ForcedDecoder:
    """The device-resident decode loop for B sequences: state records + token buffers + one hipGraph per iteration.

    load() uploads the conditioning (context tokens, chord progressions, uniform variates), run() replays the
    iteration graph until every sequence is finished, sequences() downloads the results."""

    POLL = 16          # iterations between two looks at the `done` flags (one D2H of B ints)

    def __init__(self, model, B: int, generation_length: int, memory_length: int, temperature: float, top_k: int,
                 max_chords: int = 64, record_trace: bool = False, top_p: float = 1.0):
        self.model, self.B = model, B
        self.generation_length, self.temperature, self.top_k = int(generation_length), float(temperature), int(top_k)
        self.top_p = float(top_p)          # nucleus filter after top-k (extra mode; 1.0 = the reference's behaviour)
        dev = next(model.parameters()).device
        self.dev = dev
        self.NF = call("commu_forcing_state_ints")
        self.n_ctx_max = 16
        # a sequence grows by at most one token per iteration; its cache by at most one row per iteration
        lmax = min(int(memory_length) + 1, DecodeState.MAX_POSITIONS)
        if self.n_ctx_max + self.generation_length + 1 > lmax:
            raise CommuHipError(
                f"context + generation_length ({self.generation_length}) exceeds the decode memory of {lmax} positions; the "
                "reference would start sliding its memory window here, which the K/V-cache step does not implement")
        self.state = DecodeState(model, B, lmax)
        self.ld_seq = self.n_ctx_max + self.generation_length + 2
        self.ld_chord = max_chords
        self.ld_u = self.generation_length + 1
        i32, u8 = torch.int32, torch.uint8
        self.fsm = torch.zeros(B, self.NF, dtype=i32, device=dev)
        self.seq = torch.zeros(B, self.ld_seq, dtype=i32, device=dev)
        self.chord_tok = torch.zeros(B, max_chords, dtype=i32, device=dev)
        self.chord_pos = torch.zeros(B, max_chords, dtype=i32, device=dev)
        self.wrong = torch.zeros(B, TOKEN_OFFSET.VOCAB_SIZE, dtype=u8, device=dev)
        self.utable = torch.full((B, self.ld_u), 0.5, dtype=F32, device=dev)
        self.tok = torch.zeros(B, dtype=torch.long, device=dev)
        self.active = torch.zeros(B, dtype=u8, device=dev)
        self.keep = torch.zeros(B, dtype=u8, device=dev)
        self.draw = torch.zeros(B, dtype=u8, device=dev)
        self.uni = torch.zeros(B, dtype=F32, device=dev)
        self.token = torch.zeros(B, dtype=i32, device=dev)
        self.probs = None
        self.ld_trace = 2 * (self.generation_length + 2) if record_trace else 0
        self.trace = torch.zeros(B, self.ld_trace, dtype=i32, device=dev) if record_trace else None
        self.graph = None
        self.graph_long = None
        self.n_cond = 0

    # ---- one loop iteration = decide (pre) -> model step -> sampling step -> book-keeping (post), as kernel launches
    # on the current stream.  The captured graph holds [step, {sample, post, pre of the NEXT iteration} as one launch]
    # (body_pre): run() issues the very first `pre` on its own and the kernel sequence is the same captured or not.
    # pre() / body() are the separate launches (iteration(): tests look at the draws between the stages).
    def pre(self):
        call("commu_forcing_pre", _p(self.fsm), _p(self.seq), self.ld_seq, _p(self.chord_tok), _p(self.chord_pos),
             self.ld_chord, _p(self.wrong), _p(self.utable), self.ld_u, self.generation_length, _p(self.tok),
             _p(self.active), _p(self.keep), _p(self.draw), _p(self.uni), _p(self.trace), self.ld_trace, self.B, _s())

    def body(self, want_probs: bool = False):
        B = self.B
        self.state.step(self.tok, self.active, None)
        if want_probs and self.probs is None:
            self.probs = torch.zeros(B, TOKEN_OFFSET.VOCAB_SIZE, device=self.dev)
        ops.sample_topk(self.state.logits, self.temperature, self.top_k, wrong=self.wrong, uniforms=self.uni,
                        active=self.draw, token=self.token, probs_out=self.probs if want_probs else None,
                        top_p=self.top_p)
        call("commu_forcing_post", _p(self.fsm), _p(self.seq), self.ld_seq, _p(self.chord_pos), self.ld_chord,
             _p(self.wrong), _p(self.draw), _p(self.token), None, _p(self.state.klen), _p(self.keep), self.state.Lmax,
             B, _s())

    def body_pre(self):
        """body() followed by pre() with the three per-sequence stages (sampling step, post, pre) as one launch: what
        run() issues per iteration, captured or not."""
        st = self.state
        st.step(self.tok, self.active, None)
        call("commu_decode_sample_post_pre", _p(st.logits), st.logits.stride(0), TOKEN_OFFSET.VOCAB_SIZE, _p(self.wrong),
             self.temperature, self.top_k, self.top_p, _p(self.token), None, 0, _p(self.fsm), _p(self.seq), self.ld_seq,
             _p(self.chord_tok), _p(self.chord_pos), self.ld_chord, _p(self.utable), self.ld_u, self.generation_length,
             _p(self.tok), _p(self.active), _p(self.keep), _p(self.draw), _p(self.uni), _p(self.trace), self.ld_trace,
             _p(st.klen), st.Lmax, self.B, _s())

    def iteration(self, want_probs: bool = False):
        """One complete iteration, eagerly (tests inspect the draws between iterations)."""
        self.pre()
        self.body(want_probs)

    # The split-key iteration graph pays off when FEW sequences are still alive at LONG memories (one workgroup per
    # (sequence, head) pair then streams ~1 MB alone while most of the chip idles); with all 64 sequences alive the plain
    # kernel already fills the chip and the in-launch combine only costs (measured: 0.60 against 0.37 ms per iteration).
    # (tests/probes/decode_long_rows.py, 64 sequences to completion: no split 77.8 k tokens/s; rows <= 16 / 24 / 32 / 48 with
    #  8 splits 88.9 / 93.8 / 89.2 / 85.2 k; rows <= 32 with 4 splits 93.0 k and the 256-sequence stream 107 k)
    LONG_KLEN = 768        # memories beyond this ...
    LONG_ROWS = 32         # ... and at most this many live sequences
    LONG_SPLITS = 4

    def build_graph(self, long: bool = False):
        """Capture [body, pre].  The warm-up run that the capture needs (allocator pools, lazy module state) is made
        on a scratch copy of every buffer the iteration mutates, which is restored afterwards.  long: the variant for
        long memories (commu_decode_attn_split: up to LONG_SPLITS workgroups per (sequence, head) pair)."""
        st = self.state
        # (the K/V rows the warm-up appends at klen are rewritten by the real run: the caches need no copy)
        bufs = (self.fsm, self.seq, self.wrong, st.klen, st.logits, self.tok, self.active, self.keep, self.draw, self.uni)
        saved = [t.clone() for t in bufs]
        tr = None if self.trace is None else self.trace.clone()
        keep_splits, st.attn_splits = st.attn_splits, (self.LONG_SPLITS if long else 1)
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                self.body_pre()
            torch.cuda.current_stream().wait_stream(side)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self.body_pre()
        finally:
            st.attn_splits = keep_splits
        for dst, src in zip(bufs, saved):
            dst.copy_(src)
        if tr is not None:
            self.trace.copy_(tr)
        if long:
            self.graph_long = g
        else:
            self.graph = g
        return g

    def load(self, encoded_metas: Sequence[Sequence[int]], input_datas, uniforms: Optional[np.ndarray] = None):
        """Conditioning of all B sequences: context [0] + meta[:n-1] into the caches (the n-th meta token is fed by
        the first loop iteration, its memory discarded: quirk Q3), chord progressions, variates."""
        B = self.B
        assert len(encoded_metas) == B and len(input_datas) == B
        n_cond = len(encoded_metas[0])
        if n_cond + 1 > self.n_ctx_max or any(len(m) != n_cond for m in encoded_metas):
            raise CommuHipError("encoded_meta: every sequence needs the same number (<= 15) of conditioning tokens")
        self.n_cond = n_cond
        self._last_load = (encoded_metas, input_datas, uniforms)
        if getattr(self.state, "t_err", None) is not None:
            self.state.t_err.zero_()          # (a hand-off timeout of an earlier request must not fail this one)
        ctx = torch.tensor([[0] + list(m[:n_cond - 1]) for m in encoded_metas], dtype=torch.long).t().contiguous()
        self.state.kc.zero_()
        self.state.vc.zero_()
        self.state.repack()          # (the model may have been trained since the decoder was built)
        self.state.prefill(ctx.to(self.dev))
        fsm = np.zeros((B, self.NF), dtype=np.int32)
        seq = np.zeros((B, self.ld_seq), dtype=np.int32)
        ctok = np.zeros((B, self.ld_chord), dtype=np.int32)
        cpos = np.zeros((B, self.ld_chord), dtype=np.int32)
        self.reports: List[ForcingReport] = []
        for b, (meta, data) in enumerate(zip(encoded_metas, input_datas)):
            comps = data.chord_token_components
            ct, cp = list(comps["chord_token"]), list(comps["chord_position"])
            if len(ct) != len(cp):
                raise AssertionError("Wrong Chord Length")                      # midi_inferrer.py:27
            if len(ct) > self.ld_chord:
                raise CommuHipError(f"{len(ct)} chords > max_chords={self.ld_chord}")
            seq[b, 0] = 0
            seq[b, 1:1 + n_cond] = meta[:n_cond]
            ctok[b, :len(ct)], cpos[b, :len(cp)] = ct, cp
            nm = float(data.num_measures)
            rep = ForcingReport(len(ct), nm)
            self.reports.append(rep)
            # record: len, forced, redo, first, filled, done, failed, iters, nbar, nchord, cur, length_fit, ndraw, ntrace
            fsm[b] = [1 + n_cond, -1, 0, 1, int(nm % 4 == 0), 0, 0, 0, 0, len(ct), 0, int(rep.length_fit), 0, 0]
        self.fsm.copy_(torch.from_numpy(fsm))
        self.seq.copy_(torch.from_numpy(seq))
        self.chord_tok.copy_(torch.from_numpy(ctok))
        self.chord_pos.copy_(torch.from_numpy(cpos))
        self.wrong.zero_()
        if self.trace is not None:
            self.trace.zero_()
        if uniforms is not None:
            u = np.full((B, self.ld_u), 0.5, dtype=np.float32)
            n = min(uniforms.shape[1], self.ld_u)
            u[:, :n] = uniforms[:, :n]
            self.utable.copy_(torch.from_numpy(u))

    # ---- slots: a finished sequence's slot can be re-armed for another attempt of the SAME request (same conditioning
    # tokens and chords) without touching the other slots: the context rows of its K/V cache are what they were, so only
    # the state record, the lengths, the rejected-token map and the variates are reset.  The re-armed slot sits out the
    # iteration in flight (its decision was taken from the finished record) and starts with the next one.
    def rearm(self, b: int, uniforms_row: Optional[np.ndarray] = None):
        rep0 = self.reports[b]
        rep = ForcingReport(rep0.n_chords, rep0.num_measures)
        self.reports[b] = rep
        rec = [1 + self.n_cond, -1, 0, 1, int(rep.num_measures % 4 == 0), 0, 0, 0, 0, rep.n_chords, 0,
               int(rep.length_fit), 0, 0]
        self.fsm[b].copy_(torch.tensor(rec, dtype=torch.int32))
        self.wrong[b].zero_()
        self.state.klen[b] = self.n_cond
        if uniforms_row is not None:
            u = np.full(self.ld_u, 0.5, dtype=np.float32)
            n = min(len(uniforms_row), self.ld_u)
            u[:n] = uniforms_row[:n]
            self.utable[b].copy_(torch.from_numpy(u))
        if self.trace is not None:
            self.trace[b].zero_()

    def harvest(self, b: int, fsm_row):
        """Token list of slot b (None where nothing could be drawn, Q12) from its downloaded state record."""
        self.reports[b].consumed = int(fsm_row[10])
        if fsm_row[6]:
            return None
        return self.seq[b, :int(fsm_row[0])].cpu().tolist()

    # ---- done flags without stalling the GPU: the flags of window k are copied to pinned memory behind window k and
    # looked at after window k + 1 has been queued, so the device never waits for the host between windows (a
    # synchronous read costs the wake-up + first launch, ~60 us per 16 iterations); the price is that up to POLL
    # iterations are queued beyond the one that finished the last sequence (they do nothing: every slot is inactive)
    def poll_submit(self):
        if getattr(self, "_done_pin", None) is None:
            self._done_pin = torch.zeros(self.B, self.NF, dtype=torch.int32).pin_memory()
            self._done_ev = torch.cuda.Event()
        self._done_pin.copy_(self.fsm, non_blocking=True)
        self._done_ev.record()

    def poll_result(self):
        """The state records as of the last poll_submit (numpy [B, NF]; waits for that copy only)."""
        self._done_ev.synchronize()
        return self._done_pin.numpy().copy()

    def run_iterations(self, n: int, use_graph: bool = True, klen_bound: int = 0, live_rows: Optional[int] = None):
        """n iterations.  klen_bound: an upper bound of the memory lengths during them (0: unknown / short), live_rows: the
        number of sequences still decoding (None: unknown = all): beyond LONG_KLEN with at most LONG_ROWS live sequences
        the split-key iteration graph runs (built on first use)."""
        long = klen_bound > self.LONG_KLEN and live_rows is not None and live_rows <= self.LONG_ROWS
        if use_graph and long and getattr(self, "graph_long", None) is None:
            self.build_graph(long=True)
        g = self.graph_long if (use_graph and long) else self.graph
        if not use_graph:
            keep_splits, self.state.attn_splits = self.state.attn_splits, (self.LONG_SPLITS if long else 1)
        try:
            for _ in range(n):
                if use_graph:
                    g.replay()
                else:
                    self.body_pre()
        finally:
            if not use_graph:
                self.state.attn_splits = keep_splits

    def run(self, use_graph: bool = True):
        try:
            self._run(use_graph)
        except CommuHipError:
            # a layer-tail launch gave up at a hand-off (e.g. the GPU was shared and its workgroups were not co-resident):
            # generate the request again on the chain of per-Linear launches instead of failing it
            if not self.state.tail_ok or getattr(self, "_last_load", None) is None:
                raise
            self.state.tail_ok = False
            self.graph = self.graph_long = None
            self.load(*self._last_load)
            self._run(use_graph)

    def _run(self, use_graph: bool = True):
        if use_graph and self.graph is None:
            self.build_graph()
        self.pre()                                          # decision of the first iteration
        it, pending, live = 0, False, None
        while it < self.generation_length + 1:
            self.run_iterations(self.POLL, use_graph, klen_bound=self.n_cond + it + self.POLL, live_rows=live)
            it += self.POLL
            if pending:
                rec = self.poll_result()                                  # the records one window back: no stall
                if bool(rec[:, 5].all()):
                    break
                live = int((rec[:, 5] == 0).sum())
            self.poll_submit()
            pending = True
        torch.cuda.current_stream().synchronize()
        self.state.check()

    def sequences(self):
        """Token lists (None where nothing could be drawn, Q12) and, per sequence, the model-step trace
        [(fed token, memory length before, after)] when it was recorded."""
        fsm = self.fsm.cpu().numpy()
        seq = self.seq.cpu().numpy()
        out, traces = [], []
        tr = None if self.trace is None else self.trace.cpu().numpy()
        for b in range(self.B):
            self.reports[b].consumed = int(fsm[b, 10])
            out.append(None if fsm[b, 6] else seq[b, :fsm[b, 0]].tolist())
            if tr is not None:
                klen, t = self.n_cond, []
                for k in range(min(int(fsm[b, 13]), self.ld_trace // 2)):
                    t.append((int(tr[b, 2 * k]), klen, klen + 1))
                    klen += int(tr[b, 2 * k + 1])
                traces.append(t)
        return out, traces


class BatchedGenerator:
    """Generates `len(input_datas)` sequences in parallel (temperature / top_k shared by the batch, per-sequence
    `num_measures` / `chord_token_components`).  Decoders (caches + captured graph) are kept per batch size."""

    def __init__(self, model, device, generation_length=4096, memory_length=4146):
        self.model, self.device = model, device
        self.generation_length, self.memory_length = generation_length, memory_length
        self.uniform_sources = None      # optional list of callables, one per sequence (tests inject fixture variates)
        self.trace = None                # set to a list to receive per-sequence model-step traces
        self.use_graph = True
        self._decoders = {}

    def decoder(self, B, temperature, top_k, max_chords, top_p=1.0):
        key = (B, float(temperature), int(top_k), float(top_p), self.trace is not None)
        dec = self._decoders.get(key)
        if dec is None or dec.ld_chord < max_chords:
            dec = ForcedDecoder(self.model, B, self.generation_length, self.memory_length, temperature, top_k,
                                max_chords=max(64, max_chords), record_trace=self.trace is not None, top_p=top_p)
            self._decoders[key] = dec
        return dec

    @torch.no_grad()
    def generate(self, encoded_metas: Sequence[Sequence[int]], input_datas, temperature: float, top_k: int,
                 top_p: float = 1.0):
        B = len(input_datas)
        max_chords = max(len(d.chord_token_components["chord_token"]) for d in input_datas)
        dec = self.decoder(B, temperature, top_k, max_chords, top_p)
        uniforms = None
        if temperature != 0:
            srcs = self.uniform_sources or [np.random.RandomState(1000 + b).random_sample for b in range(B)]
            uniforms = np.array([[float(srcs[b]()) for _ in range(dec.ld_u)] for b in range(B)], dtype=np.float32)
        dec.load(encoded_metas, input_datas, uniforms)
        dec.run(use_graph=self.use_graph)
        seqs, traces = dec.sequences()
        if self.trace is not None:
            self.trace[:] = traces
        return seqs, dec.reports

    @staticmethod
    def attempt_uniforms(seed: int, attempt: int, n: int) -> np.ndarray:
        """The variates of attempt number `attempt` of a request (its own stream: an attempt's sequence does not depend on
        which slot or how many other attempts ran beside it)."""
        return np.random.RandomState((seed + 7919 * attempt) % (2 ** 32)).random_sample(n).astype(np.float32)

    @torch.no_grad()
    def generate_stream(self, encoded_meta: Sequence[int], input_data, temperature: float, top_k: int, need: int,
                        accept, top_p: float = 1.0, slots: int = 64, seed: int = 0, max_attempts: Optional[int] = None):
        """Attempts of ONE request (the reference's `while idx != num_generate` loop, midi_inferrer.py:338-354, which tries
        one sequence after the other) decoded in up to `slots` parallel slots, CONTINUOUSLY: a slot whose sequence has
        ended is handed to `accept(sequence, report) -> bool` and re-armed with the next attempt while the other slots keep
        decoding -- the batch does not thin out towards the end of a round.  Returns (the first `need` accepted attempts IN
        ATTEMPT ORDER -- exactly what the reference's sequential loop would return for the same per-attempt variates,
        whatever the number of slots --, attempts started); fewer when `max_attempts` attempts did not yield `need`.
        Attempt a draws from attempt_uniforms(seed, a, .) whichever slot decodes it."""
        B = max(1, min(int(slots), int(need)))
        max_chords = len(input_data.chord_token_components["chord_token"])
        dec = self.decoder(B, temperature, top_k, max_chords, top_p)
        started = B
        uni = np.stack([self.attempt_uniforms(seed, a, dec.ld_u) for a in range(B)]) if temperature != 0 else None
        dec.load([list(encoded_meta)] * B, [input_data] * B, uni)
        if self.use_graph and dec.graph is None:
            dec.build_graph()
        dec.pre()
        slot_attempt = list(range(B))          # attempt number decoded in each slot; -1: slot retired
        results = {}                           # attempt -> its sequence if accepted, False if rejected
        n_ok, out = 0, None
        dec.run_iterations(dec.POLL, self.use_graph)
        kb, nlive = 0, None                    # bound of the memory lengths, live slots (from the polled records)
        while out is None and any(a >= 0 for a in slot_attempt):
            dec.poll_submit()                  # the records after the window just queued ...
            dec.run_iterations(dec.POLL, self.use_graph, klen_bound=kb, live_rows=nlive)      # ... are read while the next window runs
            fsm = dec.poll_result()
            kb = int(fsm[:, 0].max()) + 3 * dec.POLL
            nlive = sum(1 for b_ in range(B) if slot_attempt[b_] >= 0 and not fsm[b_, 5]) + 1
            for b in range(B):
                a = slot_attempt[b]
                if a < 0 or not fsm[b, 5]:
                    continue
                seq = dec.harvest(b, fsm[b])
                ok = bool(accept(seq, dec.reports[b]))
                results[a] = seq if ok else False
                n_ok += ok
                # no further attempts once enough were accepted: only the lower-numbered ones still in flight can matter
                if n_ok >= need or (max_attempts is not None and started >= max_attempts):
                    slot_attempt[b] = -1
                    continue
                dec.rearm(b, self.attempt_uniforms(seed, started, dec.ld_u) if temperature != 0 else None)
                slot_attempt[b] = started
                started += 1
            # the answer: the first `need` accepted attempts IN ATTEMPT ORDER, known once every attempt before the last of
            # them has ended (taking them in order of completion instead would favour short sequences)
            got, a = [], 0
            while a in results and len(got) < need:
                if results[a] is not False:
                    got.append(results[a])
                a += 1
            if len(got) >= need:
                out = got
        if out is None:                        # max_attempts exhausted: whatever was accepted, in attempt order
            out = [results[a] for a in sorted(results) if results[a] is not False][:need]
        try:
            dec.state.check()
        except CommuHipError:
            dec.state.tail_ok, dec.graph, dec.graph_long = False, None, None          # later requests on this decoder: per-Linear launches
            dec.state.t_err.zero_()
            raise
        return out, started
