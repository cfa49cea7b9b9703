"""MemTransformerLM on MI355X: the reference's class API over hand-written HIP kernels.

Mirrors `commu.model.model.MemTransformerLM` (reference commu/model/model.py:423-693): same
constructor `(cfg, vocab)`, same `forward / forward_generate / reset_length / init_mems`, same
sub-module and state_dict names (so reference checkpoints load and `model.apply(weights_init)`
works), but the arithmetic of a whole forward (and its backward) is ONE schedule of kernel
launches from libcommu_hip.so -- no per-op autograd tape, activations in bf16, fp32 master
weights / LayerNorm statistics / loss, flat parameter + gradient buffers.

There is no CPU path: calling the model with CPU tensors raises.
"""
from __future__ import annotations

import math
from typing import List, Optional

import torch
import torch.nn as nn

from .. import ops
from .._lib import CommuHipError

BF16, F32 = torch.bfloat16, torch.float32
VPAD = 768          # logits leading dimension (729 -> 768)


def _no_direct_forward(self, *a, **kw):
    raise CommuHipError(f"{type(self).__name__} is a parameter holder; call MemTransformerLM.forward")


class AdaptiveEmbedding(nn.Module):
    """Parameter holder with the reference's names (model.py:380-407); d_proj == d_embed only."""

    def __init__(self, n_token, d_embed, d_proj):
        super().__init__()
        if d_proj != d_embed:
            raise CommuHipError("d_proj != d_embed is dead code in the reference and is not built")
        self.n_token, self.d_embed, self.d_proj = n_token, d_embed, d_proj
        self.cutoffs = [n_token]
        self.emb_scale = d_proj ** 0.5
        self.emb_layers = nn.ModuleList([nn.Embedding(n_token, d_embed, sparse=False)])
        self.emb_projs = nn.ParameterList()

    forward = _no_direct_forward


class ProjectedAdaptiveLogSoftmax(nn.Module):
    """Parameter holder for the output layer (model.py:6-42), n_clusters == 0 branch only."""

    def __init__(self, n_token, d_embed, d_proj):
        super().__init__()
        self.n_token, self.d_embed, self.d_proj = n_token, d_embed, d_proj
        self.cutoffs = [n_token]
        self.cutoff_ends = [0, n_token]
        self.shortlist_size = n_token
        self.n_clusters = 0
        self.head_size = n_token
        self.out_layers = nn.ModuleList([nn.Linear(d_embed, n_token)])
        self.out_projs = nn.ParameterList()

    forward = _no_direct_forward


class PositionalEmbedding(nn.Module):
    def __init__(self, demb):
        super().__init__()
        self.demb = demb
        inv_freq = 1 / (10000 ** (torch.arange(0.0, demb, 2.0) / demb))     # model.py:142
        self.register_buffer("inv_freq", inv_freq)

    forward = _no_direct_forward


class PositionwiseFF(nn.Module):
    def __init__(self, d_model, d_inner, dropout, activation="relu"):
        super().__init__()
        self.d_model, self.d_inner, self.dropout = d_model, d_inner, dropout
        # (activation "gelu": a non-default option, MemTransformerLM.ffn_activation; same module slots / state_dict names)
        act = nn.GELU() if activation == "gelu" else nn.ReLU(inplace=True)
        self.CoreNet = nn.Sequential(nn.Linear(d_model, d_inner), act, nn.Dropout(dropout),
                                     nn.Linear(d_inner, d_model), nn.Dropout(dropout))   # model.py:163-169
        self.layer_norm = nn.LayerNorm(d_model)

    forward = _no_direct_forward


class RelPartialLearnableMultiHeadAttn(nn.Module):
    def __init__(self, n_head, d_model, d_head, dropout, dropatt=0, tgt_len=None, mem_len=None):
        super().__init__()
        self.n_head, self.d_model, self.d_head, self.dropout = n_head, d_model, d_head, dropout
        self.qkv_net = nn.Linear(d_model, 3 * n_head * d_head, bias=False)     # model.py:205
        self.drop = nn.Dropout(dropout)
        self.dropatt = nn.Dropout(dropatt)
        self.o_net = nn.Linear(n_head * d_head, d_model, bias=False)           # model.py:212
        self.layer_norm = nn.LayerNorm(d_model)
        self.scale = 1 / (d_head ** 0.5)
        self.r_net = nn.Linear(d_model, n_head * d_head, bias=False)           # model.py:278

    forward = _no_direct_forward


class RelPartialLearnableDecoderLayer(nn.Module):
    def __init__(self, n_head, d_model, d_head, d_inner, dropout, **kwargs):
        super().__init__()
        self.dec_attn = RelPartialLearnableMultiHeadAttn(n_head, d_model, d_head, dropout, **kwargs)
        self.pos_ff = PositionwiseFF(d_model, d_inner, dropout)

    forward = _no_direct_forward


def _low_priority_stream(dev):
    """Side stream for work off the critical path (weight gradients, bias / LayerNorm-parameter reductions).  (This
    device offers stream priorities -1 and 0 only -- torch clamps anything else -- so the side stream cannot be made
    LOWER than the default stream the step runs on; measured gain of the side stream at the bench shape: ~1.5 %.)"""
    return torch.cuda.Stream(device=dev)


class _Saved:
    """Activations of one forward call kept for its backward."""
    __slots__ = ("T", "M", "B", "tokens", "target", "reset", "h", "cat", "qkv", "rd", "vec", "lse", "qs", "z1", "mu1",
                 "rs1", "a", "hid", "hbits", "zff", "z2", "mu2", "rs2", "pd", "hL", "logits", "ce_lse", "same_length", "mem_len",
                 "p", "patt", "seed")


class _XLLoss(torch.autograd.Function):
    """loss[T,B] = NLL of the whole network; backward runs the hand-written backward schedule."""

    @staticmethod
    def forward(ctx, model, data, target, reset, mems, *params):
        loss, new_mems, saved = model._run_forward(data, target, reset, mems, need_grad=True)
        ctx.model, ctx.saved = model, saved
        # (without this autograd hands backward() a ZERO tensor of new_mems' shape -- [L+1, M, B, D]: a 0.47-GB fill per step
        #  at the bench shape with mem_len 1024 -- for an output that is marked non-differentiable)
        ctx.set_materialize_grads(False)
        if new_mems is None:
            new_mems = torch.empty(0, device=data.device)
        ctx.mark_non_differentiable(new_mems)
        return loss, new_mems

    @staticmethod
    def backward(ctx, dloss, _dmems):
        model, saved = ctx.model, ctx.saved
        ctx.saved = None
        if saved is None:
            raise CommuHipError("backward called twice on the same forward")
        if dloss is None:          # (only new_mems was used downstream: nothing to differentiate)
            return (None,) * (5 + len(model._flat["params"]))
        grads = model._run_backward(saved, dloss.contiguous())
        return (None, None, None, None, None) + grads


class MemTransformerLM(nn.Module):
    def __init__(self, cfg, vocab):
        n_layer = cfg.MODEL.num_layers
        n_head = cfg.MODEL.num_heads
        d_model = cfg.MODEL.units
        d_head = cfg.MODEL.units // cfg.MODEL.num_heads
        d_inner = cfg.MODEL.inner_size
        dropout = cfg.MODEL.dropout
        dropatt = cfg.MODEL.attention_dropout
        tgt_len = cfg.TRAIN.tgt_length
        mem_len = cfg.TRAIN.mem_length
        super().__init__()
        self.cfg = cfg
        self.n_token = len(vocab)
        self.d_embed = d_model
        self.d_model, self.n_head, self.d_head, self.d_inner = d_model, n_head, d_head, d_inner
        self.word_emb = AdaptiveEmbedding(self.n_token, d_model, d_model)
        self.drop = nn.Dropout(dropout)
        self.n_layer = n_layer
        self.tgt_len, self.mem_len = tgt_len, mem_len
        self.max_klen = tgt_len + mem_len
        self.layers = nn.ModuleList([
            RelPartialLearnableDecoderLayer(n_head, d_model, d_head, d_inner, dropout, tgt_len=tgt_len,
                                            mem_len=mem_len, dropatt=dropatt) for _ in range(n_layer)])
        self.crit = ProjectedAdaptiveLogSoftmax(self.n_token, d_model, d_model)
        for i in range(len(self.crit.out_layers)):                     # weight tying, model.py:480-481
            self.crit.out_layers[i].weight = self.word_emb.emb_layers[i].weight
        self.same_length = cfg.MODEL.same_length
        self.clamp_len = cfg.MODEL.clamp_len
        self.detach_mems_grad = True
        self.pos_emb = PositionalEmbedding(d_model)
        self.r_w_bias = nn.Parameter(torch.Tensor(n_head, d_head))      # model.py:491-492 (uninitialised)
        self.r_r_bias = nn.Parameter(torch.Tensor(n_head, d_head))
        if d_model % 4 or d_model > 1024 or d_head > 64 or d_head < 1 or self.n_token > VPAD:
            raise CommuHipError(f"unsupported shape: d_model={d_model} d_inner={d_inner} d_head={d_head} "
                                "(this build needs d_model % 4 == 0, d_model <= 1024 and d_head <= 64)")
        # Kernel-side dimensions: d_model / d_inner rounded up to 64, d_head to 32 or 64.  When they differ from the
        # model's (the released config: d_model 500, d_head 50, d_inner 1000) the bf16 weight shadows, biases and
        # every activation are ZERO-PADDED to them -- zeros contribute nothing to any contraction; the fp32 master
        # parameters, gradients and the state_dict keep the reference's shapes.
        self._Dp = (d_model + 63) // 64 * 64
        self._DIp = (d_inner + 63) // 64 * 64
        self._DHp = 32 if d_head <= 32 else 64
        self._padded = (self._Dp, self._DIp, self._DHp) != (d_model, d_inner, d_head)
        self.attn_scale = 1.0 / math.sqrt(d_head)                      # model.py:216 (true head dimension)
        # gradient delivery: "direct" writes into the flat gradient buffer that p.grad aliases;
        # "autograd" returns the gradients to autograd (compatible with torch DDP hooks).
        self.grad_mode = "direct"
        self._flat = None

    # ------------------------------------------------------------------ reference API
    def reset_length(self, tgt_len, mem_len):                            # model.py:494-496
        self.tgt_len = tgt_len
        self.mem_len = mem_len

    def init_mems(self, n_layers):                                       # model.py:498-505
        if self.mem_len > 0:
            param = self._param_list()[0]
            return torch.empty(n_layers + 1, 0, dtype=BF16, device=param.device)
        return None

    def forward(self, data, target, reset_mems, mems):                   # model.py:678-693
        if mems is None:
            mems = self.init_mems(self.n_layer)
        if getattr(self, "parity_fp32", False):
            # the reference's arithmetic (fp32 end to end, train.py:48 `amp = None`) for the LOSS of a forward pass --
            # evaluate (train.py:74-110) and any no-grad forward; there is no fp32 backward pass
            if torch.is_grad_enabled() and any(p.requires_grad for p in self._param_list()):
                raise CommuHipError("parity_fp32 is a forward-only mode: call forward under torch.no_grad() (evaluate does), "
                                    "or switch it off for training")
            if self.training and (float(self.drop.p) > 0 or (self.n_layer > 0 and float(self.layers[0].dec_attn.dropatt.p) > 0)):
                raise CommuHipError("parity_fp32 has eval-mode semantics (no dropout): call model.eval() first")
            with torch.no_grad():
                nll, new_mems, _ = self._run_forward_f32(data, mems, reset=reset_mems, target=target)
            return nll, new_mems
        if torch.is_grad_enabled() and any(p.requires_grad for p in self._param_list()):
            self._ensure_flat()
            loss, new_mems = _XLLoss.apply(self, data, target, reset_mems, mems, *self._flat["params"])
            if mems is None:
                new_mems = None
            return loss, new_mems
        loss, new_mems, _ = self._run_forward(data, target, reset_mems, mems, need_grad=False)
        return loss, new_mems

    def forward_generate(self, data, mems):                              # model.py:606-628
        if mems is None:
            mems = self.init_mems(self.n_layer)
        assert self.crit.n_clusters == 0
        if getattr(self, "parity_fp32", False):          # the reference's arithmetic: fp32 end to end (see _run_forward_f32)
            with torch.no_grad():
                logits, new_mems, _ = self._run_forward_f32(data, mems)
            return logits, new_mems
        with torch.no_grad():
            logits, new_mems, _ = self._run_forward(data, None, None, mems, need_grad=False, want_logits=True)
        return logits, new_mems

    def _run_forward_f32(self, data, mems, want_kv=False, reset=None, target=None):
        """forward_generate (model.py:606-628 -> _forward :576-604) in the reference's own arithmetic: fp32 master weights
        (no bf16 shadows, no padding), fp32 activations and memory, fp32 MFMA Linears, accurate transcendentals
        (csrc/parity_f32.hip).  The mode behind the "bit-exact greedy tokens" claim: `model.parity_fp32 = True` /
        `generate.py --parity`; what differs from the reference is summation order only.  Eval-mode semantics (no dropout),
        memory as the reference keeps it: a fp32 [L+1, M, B, d_model] tensor.
        reset (optional bool [B]): `reset_mems` of forward (model.py:558-574: that sequence does not see the memory).
        target (optional [T, B]): return the per-token NLL [T, B] (model.py:689-691, fp32 log-softmax) instead of the logits."""
        params = self._param_list()
        dev = params[0].device
        if dev.type != "cuda" or not data.is_cuda:
            raise CommuHipError("MemTransformerLM runs on an MI355X only (no CPU fallback)")
        T, B = data.shape
        D, H, DH, L, V = self.d_model, self.n_head, self.d_head, self.n_layer, self.n_token
        HD = H * DH
        M = 0 if mems is None or mems.numel() == 0 else mems.shape[1]
        if M > 0 and mems.dtype != F32:
            mems = mems.to(F32)
        if M > 0 and not mems.is_contiguous():
            mems = mems.contiguous()
        K = T + M
        E = self.word_emb.emb_layers[0].weight
        h = ops.embed_f32(data.contiguous().view(-1), E)                                   # model.py:585
        pd = ops.posemb_f32(self.pos_emb.inv_freq, K, D, clamp_len=int(self.clamp_len))    # :578-584 (by distance)
        u, vb = self.r_w_bias.contiguous(), self.r_r_bias.contiguous()
        rst = None
        if reset is not None and M > 0:
            rst = reset.to(device=dev, dtype=torch.uint8).contiguous()
        hids = [h]
        kv_out = []
        for i in range(L):
            lay = self.layers[i]
            att, ff = lay.dec_attn, lay.pos_ff
            Wqkv = att.qkv_net.weight
            qkv = torch.empty(K * B, 3 * HD, device=dev, dtype=F32)
            if M > 0:                                       # memory rows: k | v only (their q third is never used, :306)
                ops.gemm_nt_f32(mems[i].reshape(M * B, D), Wqkv[HD:], out=qkv[:M * B, HD:])
            ops.gemm_nt_f32(h, Wqkv, out=qkv[M * B:])
            rd = ops.gemm_nt_f32(pd, att.r_net.weight)                                     # :308-310
            vec = ops.relattn_f32(qkv[M * B:, :HD], qkv[:, HD:2 * HD], qkv[:, 2 * HD:], B * 3 * HD, 3 * HD, rd, u, vb,
                                  T, M, B, H, DH, bool(self.same_length), int(self.mem_len), self.attn_scale, reset=rst)
            z1 = ops.gemm_nt_f32(vec, att.o_net.weight, resid=h)                           # :344-349
            a = ops.layernorm_f32(z1, att.layer_norm.weight, att.layer_norm.bias, att.layer_norm.eps)
            hid = ops.gemm_nt_f32(a, ff.CoreNet[0].weight, bias=ff.CoreNet[0].bias, relu=True)     # :163-181
            z2 = ops.gemm_nt_f32(hid, ff.CoreNet[3].weight, bias=ff.CoreNet[3].bias, resid=a)
            h = ops.layernorm_f32(z2, ff.layer_norm.weight, ff.layer_norm.bias, ff.layer_norm.eps)
            hids.append(h)
            if want_kv:
                kv_out.append(qkv)
        # K9 (model.py:507-538): cat(mems, hids)[max(0, M + T - mem_len) : M + T]
        new_mems = None
        if mems is not None:
            hs = torch.stack([x.view(T, B, D) for x in hids])
            allm = hs if M == 0 else torch.cat([mems, hs], dim=1)
            end = M + T
            beg = max(0, end - self.mem_len)
            new_mems = allm[:, beg:end].contiguous()
        logits = ops.gemm_nt_f32(h, E, bias=self.crit.out_layers[0].bias)                  # :46,620-626 (tied weight)
        if target is not None:                                                             # :689-691 (fp32 log-softmax + gather)
            nll, _ = ops.ce_fwd(logits, target.contiguous().view(-1).to(dev), V)
            return nll.view(T, B), new_mems, (kv_out if want_kv else None)
        return logits.view(T, B, V), new_mems, (kv_out if want_kv else None)

    def zero_grad(self, set_to_none: bool = True):                      # nn.Module.zero_grad without the module-tree walk
        for p in self._param_list():
            if p.grad is not None:
                if set_to_none:
                    p.grad = None
                else:
                    p.grad.detach_().zero_()

    # ------------------------------------------------------------------ flat parameter storage
    def _param_list(self):
        # (cached: walking the module tree costs more host time per step than it looks -- the parameter OBJECTS never
        #  change after construction; .to() / load_state_dict swap or copy their data in place)
        pl = self.__dict__.get("_plist")
        if pl is None:
            pl = [p for _, p in self.named_parameters()]
            self.__dict__["_plist"] = pl
        return pl

    def _ensure_flat(self):
        """fp32 master parameters / gradients live in two flat device buffers that the individual
        nn.Parameters alias; bf16 shadows (W and W^T) are refreshed when any parameter changed."""
        params = self._param_list()
        dev = params[0].device
        if dev.type != "cuda":
            raise CommuHipError("MemTransformerLM runs on an MI355X only: move it with .to('cuda') "
                                "(there is no CPU fallback)")
        fl = self._flat
        ok = fl is not None and fl["dev"] == dev and all(
            p.data_ptr() == fl["p"].data_ptr() + 4 * off for p, off in zip(params, fl["offs"]))
        if not ok:
            offs, total = [], 0
            for p in params:
                offs.append(total)
                total += (p.numel() + 63) // 64 * 64
            flat_p = torch.zeros(total, device=dev, dtype=F32)
            flat_g = torch.zeros(total, device=dev, dtype=F32)
            for p, off in zip(params, offs):
                if p.dtype != F32:
                    raise CommuHipError("parameters must be fp32 (master weights)")
                flat_p[off:off + p.numel()].copy_(p.data.reshape(-1))
                p.data = flat_p[off:off + p.numel()].view(p.shape)
                if p.grad is not None:
                    flat_g[off:off + p.numel()].copy_(p.grad.reshape(-1))
                    p.grad = flat_g[off:off + p.numel()].view(p.shape)
            fl = {"dev": dev, "p": flat_p, "g": flat_g, "offs": offs, "params": params, "total": total,
                  "bf16": torch.zeros(total, device=dev, dtype=BF16), "version": None, "shadow": {},
                  "m": None, "v": None, "gnorm": torch.zeros(1, device=dev, dtype=F32),
                  "gpart": torch.zeros(256, device=dev, dtype=F32)}
            self._flat = fl
            self._name_off = {n: o for (n, _), o in zip(self.named_parameters(), offs)}
        ver = sum(p._version for p in params)
        if fl["version"] != ver:
            self._refresh_shadows()
            fl["version"] = sum(p._version for p in params)
        return fl

    def _bf16_view(self, name, shape):
        fl = self._flat
        off = self._name_off[name]
        n = 1
        for s in shape:
            n *= s
        return fl["bf16"][off:off + n].view(shape)

    def _grad_view(self, name, shape):
        fl = self._flat
        off = self._name_off[name]
        n = 1
        for s in shape:
            n *= s
        return fl["g"][off:off + n].view(shape)

    def _refresh_shadows(self, cast=True):
        """bf16 copy of every parameter (same offsets) + transposed bf16 weights for the dX GEMMs.
        Padded shapes (see __init__): zero-padded bf16 weights / fp32 biases rebuilt from the master copy."""
        fl = self._flat
        if cast:
            ops.cast_bf16(fl["p"], fl["bf16"])
        D, DI, HD, V = self.d_model, self.d_inner, self.n_head * self.d_head, self.n_token
        sh = fl["shadow"]
        dev = fl["dev"]
        if self._padded:
            return self._refresh_padded_shadows()

        pairs = []

        def tr(key, name, shape, pad_cols=None):
            w = self._bf16_view(name, shape)
            cols = shape[0] if pad_cols is None else pad_cols
            if key not in sh:
                sh[key] = torch.zeros(shape[1], cols, device=dev, dtype=BF16)
            pairs.append((w, sh[key][:, :shape[0]]))

        def transposes():          # every W^T shadow of the step in ONE launch (25 launches of a few dozen workgroups before)
            tr("Et", "word_emb.emb_layers.0.weight", (V, D), VPAD)
            for i in range(self.n_layer):
                pre = f"layers.{i}."
                tr(f"qkv_t{i}", pre + "dec_attn.qkv_net.weight", (3 * HD, D))
                tr(f"o_t{i}", pre + "dec_attn.o_net.weight", (D, HD))
                tr(f"w1_t{i}", pre + "pos_ff.CoreNet.0.weight", (DI, D))
                tr(f"w2_t{i}", pre + "pos_ff.CoreNet.3.weight", (D, DI))
            ops.transpose_group_bf16(pairs)
        # The transposed shadows are read by the NEXT backward only (forward and decode use the plain bf16 copy): build
        # them on the side stream, beside the next forward; _run_backward waits for fl["shadow_ready"].
        if getattr(self, "wgrad_side_stream", True):
            side = fl.get("wgrad_stream")
            if side is None:
                side = fl["wgrad_stream"] = _low_priority_stream(dev)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                side.wait_event(ev)
                transposes()
                fl["shadow_ready"] = torch.cuda.Event()
                fl["shadow_ready"].record(side)
        else:
            transposes()

    # crop / pad specification of a 2-D parameter: rows = rg groups of rt (padded to rp), cols = cg groups of ct
    # (padded to cp).  Head-structured dimensions (q|k|v thirds x heads) pad every head separately.
    def _spec(self, kind):
        D, DI, H, DH, V = self.d_model, self.d_inner, self.n_head, self.d_head, self.n_token
        Dp, DIp, DHp = self._Dp, self._DIp, self._DHp
        return {"qkv": (3 * H, DH, DHp, 1, D, Dp), "kv": (2 * H, DH, DHp, 1, D, Dp), "o": (1, D, Dp, H, DH, DHp),
                "r": (H, DH, DHp, 1, D, Dp), "w1": (1, DI, DIp, 1, D, Dp), "w2": (1, D, Dp, 1, DI, DIp),
                "E": (1, V, V, 1, D, Dp), "Egrad": (1, V, VPAD, 1, D, Dp)}[kind]

    def _refresh_padded_shadows(self):
        fl = self._flat
        sh, dev = fl["shadow"], fl["dev"]
        H, DH, DHp = self.n_head, self.d_head, self._DHp

        pmap = dict(self.named_parameters())
        tpairs = []          # (padded weight, its transposed shadow): one grouped launch at the end

        def fp32(name):
            return pmap[name].data

        def padw(key, name, kind, transpose_key=None, tcols=None):
            rg, rt, rp, cg, ct, cp = self._spec(kind)
            if key not in sh:
                sh[key] = torch.zeros(rg * rp, cg * cp, device=dev, dtype=BF16)
            sh[key].view(rg, rp, cg, cp)[:, :rt, :, :ct].copy_(fp32(name).view(rg, rt, cg, ct))
            if transpose_key is not None:
                if transpose_key not in sh:
                    sh[transpose_key] = torch.zeros(cg * cp, rg * rp if tcols is None else tcols, device=dev, dtype=BF16)
                tpairs.append((sh[key], sh[transpose_key][:, :rg * rp]))

        def padv(key, name, groups, true, pad):
            if key not in sh:
                sh[key] = torch.zeros(groups * pad, device=dev, dtype=F32)
            sh[key].view(groups, pad)[:, :true].copy_(fp32(name).reshape(groups, true))
        padw("E", "word_emb.emb_layers.0.weight", "E", "Et", VPAD)
        padv("u", "r_w_bias", H, DH, DHp)
        padv("vb", "r_r_bias", H, DH, DHp)
        for i in range(self.n_layer):
            pre = f"layers.{i}."
            padw(f"qkv{i}", pre + "dec_attn.qkv_net.weight", "qkv", f"qkv_t{i}")
            padw(f"o{i}", pre + "dec_attn.o_net.weight", "o", f"o_t{i}")
            padw(f"r{i}", pre + "dec_attn.r_net.weight", "r")
            padw(f"w1{i}", pre + "pos_ff.CoreNet.0.weight", "w1", f"w1_t{i}")
            padw(f"w2{i}", pre + "pos_ff.CoreNet.3.weight", "w2", f"w2_t{i}")
            padv(f"b1{i}", pre + "pos_ff.CoreNet.0.bias", 1, self.d_inner, self._DIp)
            padv(f"b2{i}", pre + "pos_ff.CoreNet.3.bias", 1, self.d_model, self._Dp)
        ops.transpose_group_bf16(tpairs)

    # ------------------------------------------------------------------ forward schedule
    def _weights(self, i):
        """bf16 GEMM operands and fp32 biases of layer i at the kernel-side (possibly padded) shapes."""
        D, DI, HD = self.d_model, self.d_inner, self.n_head * self.d_head
        pre = f"layers.{i}."
        lay = self.layers[i]
        if self._padded:
            sh = self._flat["shadow"]
            return {"qkv": sh[f"qkv{i}"], "o": sh[f"o{i}"], "r": sh[f"r{i}"], "w1": sh[f"w1{i}"], "w2": sh[f"w2{i}"],
                    "b1": sh[f"b1{i}"], "b2": sh[f"b2{i}"]}
        bv = self._bf16_view
        return {"qkv": bv(pre + "dec_attn.qkv_net.weight", (3 * HD, D)),
                "o": bv(pre + "dec_attn.o_net.weight", (D, HD)),
                "r": bv(pre + "dec_attn.r_net.weight", (HD, D)),
                "w1": bv(pre + "pos_ff.CoreNet.0.weight", (DI, D)),
                "w2": bv(pre + "pos_ff.CoreNet.3.weight", (D, DI)),
                "b1": lay.pos_ff.CoreNet[0].bias, "b2": lay.pos_ff.CoreNet[3].bias}

    def _weights_fp8(self, i):
        """MX-fp8 (bytes, scales) of layer i's four Linear weights, re-quantised whenever the parameters changed
        (`fp8_forward`, BASELINE.json configs[4])."""
        fl = self._flat
        cache = fl.setdefault("fp8", {})
        if cache.get("version") != fl["version"]:
            cache.clear()
            cache["version"] = fl["version"]
        if i not in cache:
            w = self._weights(i)
            cache[i] = {k: ops.quant_mxfp8(w[k]) for k in ("qkv", "o", "w1", "w2")}
        return cache[i]

    def _emb_bf16(self):
        """[V, Dp] bf16 embedding / output-layer weight."""
        if self._padded:
            return self._flat["shadow"]["E"]
        return self._bf16_view("word_emb.emb_layers.0.weight", (self.n_token, self.d_model))

    def _uv(self):
        """fp32 r_w_bias, r_r_bias as [H * DHp]."""
        if self._padded:
            sh = self._flat["shadow"]
            return sh["u"], sh["vb"]
        return self.r_w_bias, self.r_r_bias

    def _mems_in(self, mems):
        """Memory tensor [L+1, M, B, D] -> kernel-side [L+1, M, B, Dp] (zero padded)."""
        if not self._padded or mems is None or mems.numel() == 0 or mems.shape[-1] == self._Dp:
            return mems
        Lp, M, B, D = mems.shape
        if mems.dtype == BF16 and mems.stride()[1:] == (B * self._Dp, self._Dp, 1):
            return torch.as_strided(mems, (Lp, M, B, self._Dp), mems.stride(), mems.storage_offset())   # our own padded storage
        out = torch.zeros(Lp, M, B, self._Dp, device=mems.device, dtype=BF16)
        out[..., :D].copy_(mems)
        return out

    def _run_forward(self, data, target, reset, mems, need_grad, want_logits=False, want_kv=False):
        fl = self._ensure_flat()
        dev = fl["dev"]
        if not data.is_cuda:
            raise CommuHipError("inputs must be GPU tensors (no CPU fallback)")
        T, B = data.shape
        # kernel-side (possibly zero-padded) dimensions; Dt = the model's d_model (LayerNorm, embedding scale)
        Dt, D, DI, H, DH, L, V = self.d_model, self._Dp, self._DIp, self.n_head, self._DHp, self.n_layer, self.n_token
        HD = H * DH
        mems = self._mems_in(mems)
        M = 0 if mems is None or mems.numel() == 0 else mems.shape[1]
        K = T + M
        TB = T * B
        tokens = data.contiguous().view(-1)
        if getattr(self, "check_inputs", False):
            # the reference raises IndexError on ids outside the vocabulary (nn.Embedding / gather); the kernels do
            # not bounds-check, so debugging runs can turn this host-synchronising check on
            lo, hi = int(tokens.min()), int(tokens.max())
            if lo < 0 or hi >= V or (target is not None and (int(target.min()) < 0 or int(target.max()) >= V)):
                raise IndexError(f"token id out of range [0, {V}): min {lo}, max {hi}")
        rst = None
        if reset is not None and M > 0:
            rst = reset.to(device=dev, dtype=torch.uint8).contiguous()
        lay = [self.layers[i] for i in range(L)]
        sv = _Saved() if need_grad else None

        # K16 dropout: active in train() mode; one base seed per forward call, one derived seed per site
        p = float(self.drop.p) if self.training else 0.0
        patt = float(lay[0].dec_attn.dropatt.p) if (self.training and L > 0) else 0.0
        seed = 0
        if p > 0 or patt > 0:          # (`fixed_drop_seed`: tests pin the base seed; default: torch's CPU generator)
            fixed = getattr(self, "fixed_drop_seed", None)
            seed = int(fixed) if fixed is not None else int(torch.randint(0, 2 ** 31 - 1, (1,)).item())

        def ss(site):
            return ops.site_seed(seed, site)

        # every layer's output goes straight into one [L+1, T*B, Dp] buffer: it IS the list of hidden states K9 needs
        hids = torch.empty(L + 1, TB, D, device=dev, dtype=BF16)
        h = ops.embed_fwd(tokens, self.word_emb.emb_layers[0].weight, out=hids[0], drop_p=p, drop_seed=ss(0))    # K1
        pd = ops.posemb(self.pos_emb.inv_freq, K, Dt, drop_p=p, drop_seed=ss(1), ld=D,
                        clamp_len=int(self.clamp_len))                                               # K2 (distance order)
        if need_grad:
            sv.T, sv.M, sv.B, sv.tokens, sv.reset, sv.pd = T, M, B, tokens, rst, pd
            sv.same_length, sv.mem_len = bool(self.same_length), int(self.mem_len)
            sv.p, sv.patt, sv.seed = p, patt, seed
            for k in ("h", "cat", "qkv", "rd", "vec", "lse", "qs", "z1", "mu1", "rs1", "a", "hid", "hbits", "zff", "z2", "mu2", "rs2"):
                setattr(sv, k, [])
        u, vb = self._uv()
        h_out = None
        kv_out = []
        # K5 for all layers at once: r_net sees the same position table in every layer, and the bf16 copies of the layers'
        # weights sit at a constant stride in the flat buffer -> one batched launch instead of L small ones
        rd_all = None
        if L > 1 and not self._padded:
            o0, o1 = self._name_off["layers.0.dec_attn.r_net.weight"], self._name_off["layers.1.dec_attn.r_net.weight"]
            if o0 % 8 == 0 and (o1 - o0) % 8 == 0 and Dt % 8 == 0:
                rd_all = ops.gemm_nt_layers(pd, self._flat["bf16"][o0:], HD, Dt, o1 - o0, L)
        fp8 = bool(getattr(self, "fp8_forward", False))
        # FFN activation: "relu" (the reference, model.py:163-169: default and parity mode) or "gelu" (BASELINE.json's north
        # star names a GELU-FFN; model.ffn_activation = "gelu": training / evaluation passes only, bf16 path)
        gelu = getattr(self, "ffn_activation", "relu") == "gelu"
        if gelu and fp8:
            raise CommuHipError("ffn_activation 'gelu' is not available on the MX-fp8 forward path")
        if fp8 and (self._padded or D % 128 or DI % 128 or HD % 128):
            raise CommuHipError("fp8_forward needs d_model, d_inner and n_head * d_head to be multiples of 128")
        for i in range(L):
            w = self._weights(i)
            s0 = 16 + 4 * i
            qkv = torch.empty(K * B, 3 * HD, device=dev, dtype=BF16)
            cat = None
            # fp8_forward (opt-in, BASELINE.json configs[4]): the four Linear products of the layer's FORWARD in OCP MX-fp8
            # (activations quantised on the fly, weights once per optimiser step); backward stays bf16 on the saved bf16
            # activations.  Needs K % 128 == 0 for every contraction.
            f8 = self._weights_fp8(i) if fp8 else None
            if M > 0:                                                            # K4 over [mem; h] (model.py:283-288)
                cat = mems[i].reshape(M * B, D)
                if cat.dtype != BF16:
                    cat = cat.to(BF16)
                if fp8:
                    ops.linear_mxfp8(cat, (f8["qkv"][0][HD:], f8["qkv"][1][HD:]), out=qkv[:M * B, HD:])
                else:
                    ops.gemm_nt(cat, w["qkv"][HD:], out=qkv[:M * B, HD:])
            if fp8:
                ops.linear_mxfp8(h, f8["qkv"], out=qkv[M * B:])
            else:
                ops.gemm_nt(h, w["qkv"], out=qkv[M * B:])
            rd = rd_all[i] if rd_all is not None else ops.gemm_nt(pd, w["r"])    # K5
            vec, lse, qs = ops.relattn_fwd(qkv[M * B:, :HD], qkv[:, HD:2 * HD], qkv[:, 2 * HD:], rd, u, vb, rst,
                                           T, M, B, H, DH, bool(self.same_length), int(self.mem_len),
                                           save_q=need_grad, save_p=need_grad, drop_p=patt, drop_seed=ss(s0),
                                           scale=self.attn_scale)    # K6
            lin = (lambda x, k, **e: ops.linear_mxfp8(x, f8[k], **e)) if fp8 else (lambda x, k, **e: ops.gemm_nt(x, w[k], **e))
            z1 = lin(vec, "o", resid=h, drop_p=p, drop_seed=ss(s0 + 1))                         # K7
            a, mu1, rs1 = ops.layernorm_fwd(z1, lay[i].dec_attn.layer_norm.weight, lay[i].dec_attn.layer_norm.bias)
            # ReLU backward from ONE BIT per element, written by this GEMM's epilogue, instead of the bf16 activations
            # (model.relu_sign_bits, default on since round 6; bit-identical gradients, tests/test_model_gpu.py).  Alone the
            # backward GEMM drops from 123 to 84 us and this one costs 6 us more, 0.75 GB less HBM traffic per step; inside the
            # step the pair is worth 0.05 ms (15.73 -> 15.69 ms, 3 of 3 interleaved runs; in round 5, before the side-stream
            # launches were re-balanced, it measured 0.07 ms slower and stayed off).  Shapes the bit path does not take fall
            # back to the bf16 mask.
            hbits = None
            if need_grad and not fp8 and not gelu and getattr(self, "relu_sign_bits", True):
                nw = ops.signbits_words(TB, DI, D, a.stride(0), w["w1"].stride(0), DI)
                if nw > 0 and ops.signbits_words(TB, DI, D, D, D, DI) > 0:
                    hbits = torch.empty(nw, device=dev, dtype=torch.int32)
            zff = None
            if gelu:          # non-default activation: plain Linear, then the element-wise GELU + dropout pair
                zff = ops.gemm_nt(a, w["w1"], bias=w["b1"])
                hid = ops.gelu_fwd(zff, drop_p=p, drop_seed=ss(s0 + 2))
            elif hbits is not None:
                hid = ops.gemm_nt(a, w["w1"], bias=w["b1"], relu=True, drop_p=p, drop_seed=ss(s0 + 2), sign_bits_out=hbits)
            else:
                hid = lin(a, "w1", bias=w["b1"], relu=True, drop_p=p, drop_seed=ss(s0 + 2))     # K8
            z2 = lin(hid, "w2", bias=w["b2"], resid=a, drop_p=p, drop_seed=ss(s0 + 3))
            if i == L - 1 and p > 0:          # final `self.drop(core_out)` (model.py:601) as a second LN output
                h_out = torch.empty(TB, D, device=dev, dtype=BF16)
            y, mu2, rs2 = ops.layernorm_fwd(z2, lay[i].pos_ff.layer_norm.weight, lay[i].pos_ff.layer_norm.bias,
                                            y=hids[i + 1], y_drop=h_out, drop_p=p, drop_seed=ss(2))
            if need_grad:
                sv.h.append(h); sv.cat.append(cat); sv.qkv.append(qkv); sv.rd.append(rd); sv.vec.append(vec)
                sv.lse.append(lse); sv.qs.append(qs); sv.z1.append(z1); sv.mu1.append(mu1); sv.rs1.append(rs1)
                sv.a.append(a); sv.hid.append(hid); sv.hbits.append(hbits); sv.zff.append(zff); sv.z2.append(z2)
                sv.mu2.append(mu2); sv.rs2.append(rs2)
            if want_kv:
                kv_out.append(qkv)
            h = y
        if h_out is None:
            h_out = h

        new_mems = self._update_mems(hids, mems, M, T, B)                        # K9
        logits = torch.empty(TB, VPAD, device=dev, dtype=F32)                    # K10 / K14
        ops.gemm_nt(h_out, self._emb_bf16(), out=logits[:, :V], bias=self.crit.out_layers[0].bias)
        if want_logits:
            return logits.view(T, B, VPAD)[:, :, :V], new_mems, (kv_out if want_kv else None)
        tgt = target.contiguous().view(-1)
        nll, ce_lse = ops.ce_fwd(logits, tgt, V)
        if need_grad:
            sv.hL, sv.logits, sv.ce_lse, sv.target = h_out, logits, ce_lse, tgt
            if getattr(self, "keep_saved", False):       # tests read the saved activations (e.g. the ReLU gates) of the last call
                self.__dict__["last_saved"] = sv
        return nll.view(T, B), new_mems, sv

    def _update_mems(self, hids, mems, M, T, B):                                 # model.py:507-538
        """K9.  hids: [L+1, T*B, Dp] (written in place by the forward pass).  new_mems[l] = cat(mems[l], hids[l])[beg:end];
        with tgt_len >= mem_len that is a VIEW of hids (no copy), otherwise one commu_mems_update launch."""
        if mems is None:
            return None
        end = M + T
        beg = max(0, end - self.mem_len)
        n = end - beg
        Lp, Dp = hids.shape[0], self._Dp
        if beg >= M:
            out = hids.view(Lp, T, B, Dp)[:, beg - M:]
        else:
            if mems.dtype != BF16 or mems.stride()[1:] != (B * Dp, Dp, 1):
                mems = mems.to(BF16).contiguous()
            out = torch.empty(Lp, n, B, Dp, device=hids.device, dtype=BF16)
            ops.mems_update(hids, mems, out, beg)
        # (padded shapes: the caller sees the reference's [L+1, n, B, d_model]; the view keeps the padded storage)
        return out[..., :self.d_model] if self._padded else out

    # ------------------------------------------------------------------ backward schedule
    def _run_backward(self, sv, dloss):
        fl = self._ensure_flat()
        dev = fl["dev"]
        params = fl["params"]
        direct = self.grad_mode == "direct"
        if direct:
            fresh = all(p.grad is None for p in params)
            aliased = all(p.grad is not None and p.grad.data_ptr() == fl["g"].data_ptr() + 4 * off
                          for p, off in zip(params, fl["offs"]))
            if fresh:
                fl["g"].zero_()
                for p, off in zip(params, fl["offs"]):
                    p.grad = fl["g"][off:off + p.numel()].view(p.shape)
            elif not aliased:
                direct = False
        G = fl["g"]
        if not direct:
            G = torch.zeros_like(fl["g"])
        gname = {n: o for n, o in self._name_off.items()}

        def gv(name, shape):
            n = 1
            for s in shape:
                n *= s
            return G[gname[name]:gname[name] + n].view(shape)

        # Weight-gradient GEMMs (dW = dY^T X) are off the critical path of backward: they only feed the optimiser.
        # They run on a SIDE stream, ordered after the kernels that produced their operands, so they fill the
        # tails of the attention / dX kernels on the main stream; joined before the gradients are consumed.
        main = torch.cuda.current_stream()
        if fl.get("shadow_ready") is not None:
            main.wait_event(fl["shadow_ready"])          # transposed weights of the last optimiser step are in place
        side = fl.get("wgrad_stream") if getattr(self, "wgrad_side_stream", True) else None
        if side is None and getattr(self, "wgrad_side_stream", True):
            side = fl["wgrad_stream"] = _low_priority_stream(dev)
        keep = []          # operands stay referenced until the join: the allocator must not hand them out early

        # Weight gradients of a layer are collected and issued as ONE grouped launch of the eight-phase TN kernel
        # (commu_gemm_tn_bf16_grouped: every token slice on its own XCD, so the layer's activations leave HBM once);
        # problems the grouped kernel does not take (small shapes) fall back to the per-GEMM path.
        pending = []

        def wgrad(dY, Xa, gW, rows=None, crop=None, colsum=None):
            # colsum = (fp32 vector, n): vector[:n] += column sums of dY[:, :n] (a bias gradient), from the same launch
            pending.append((dY, Xa, gW, rows, crop, colsum))

        # Work deferred to the first side stream is collected and handed over at the next flush_wgrads() with ONE event: an
        # event record costs the main queue 6-8 us (the trace shows the gap in front of the next kernel), and a layer used to
        # record one per deferred call -- four between the band pass and the next dX GEMM.  Nothing is launched on the main
        # stream between a defer() and the flush that follows it, so the later hand-over reads the same data.
        side_fns = []

        def flush_wgrads(wide=False):
            # wide: the launch may take every CU (the last one of the pass: the main stream has one GEMM and the embedding
            # scatter left and then WAITS for it -- at half the chip it ended 0.25 ms after them, round-6 trace)
            if not pending and not side_fns:
                return
            batch = list(pending)
            pending.clear()
            fns = list(side_fns)
            side_fns.clear()
            keep.extend(t for it in batch for t in it[:2])

            def run():
                for fn in fns:
                    fn()
                groups = {}
                for it in batch:
                    groups.setdefault(it[0].shape[0], []).append(it)
                for items in groups.values():
                    self._tn_group_acc(items, budget=256 if wide else None)
            if side is None:
                return run()
            ev = torch.cuda.Event()
            ev.record(main)
            with torch.cuda.stream(side):
                side.wait_event(ev)
                run()

        # a second side stream takes the light reductions (bias / LayerNorm-parameter column sums): they are independent of
        # the weight-gradient GEMMs, and with both on one stream the main stream waited ~0.4 ms at the end of every
        # backward pass for the tail of that queue
        side2 = None
        if side is not None:
            side2 = fl.get("reduce_stream")
            if side2 is None:
                side2 = fl["reduce_stream"] = torch.cuda.Stream(device=dev)

        def join():
            if side is not None:
                main.wait_stream(side)
                main.wait_stream(side2)

        def defer_light(fn):          # like defer, on the second side stream
            if side2 is None:
                return fn()
            ev = torch.cuda.Event()
            ev.record(main)
            with torch.cuda.stream(side2):
                side2.wait_event(ev)
                fn()

        def defer(fn):          # fn's kernels go to the side stream at the next flush_wgrads() (see side_fns)
            if side is None:
                return fn()
            side_fns.append(fn)
        # dS-by-distance scratch: two buffers alternate between layers, so that the deferred dRd GEMM of layer i may
        # still read its buffer while layer i-1's attention backward fills the other one
        scr = fl.setdefault("attn_scratch", [{}, {}])
        scr_free = [None, None]          # event: the side stream is done with that buffer

        T, M, B = sv.T, sv.M, sv.B
        # Dt/DIt/DHt: the model's (state_dict) dimensions; D/DI/DH: kernel-side, possibly zero-padded
        Dt, DIt, DHt = self.d_model, self.d_inner, self.d_head
        D, DI, H, DH, L, V = self._Dp, self._DIp, self.n_head, self._DHp, self.n_layer, self.n_token
        HD, HDt, K, TB = H * DH, H * DHt, T + M, T * B
        pad = self._padded
        spec = self._spec if pad else (lambda kind: None)
        sh = fl["shadow"]
        g = dloss.reshape(-1).to(F32)
        dlogits = ops.ce_bwd(sv.logits, sv.target, sv.ce_lse, g, V)             # [TB, 768] bf16, pad cols 0
        keep.append(dlogits)
        # every final column-sum pass of the step (bias, LayerNorm-parameter, r_w_bias / r_r_bias gradients) runs as ONE
        # launch at the end of the pass; only the slab passes over the large bf16 operands are issued where they arise
        # (not under an overlapped gradient exchange: a layer's slice must be final when its hook fires)
        cgrp = ops.ColsumGroup() if getattr(self, "grad_ready_hook", None) is None else None
        keep.append(cgrp)          # its sources live until the streams are joined at the end of the pass
        # (a reduction that only joins the group launches nothing: no stream hand-over, no event)
        light = (lambda fn: fn()) if cgrp is not None else defer_light
        gE = gv("word_emb.emb_layers.0.weight", (V, Dt))
        # (the output bias gradient = column sums of dlogits: left in the slabs by the weight-gradient launch)
        wgrad(dlogits, sv.hL, gE, rows=V, crop=spec("Egrad"), colsum=(gv("crit.out_layers.0.bias", (V,)), V))
        p, patt = sv.p, sv.patt
        inv_keep = 1.0 / (1.0 - p)

        def ss(site):
            return ops.site_seed(sv.seed, site)

        # the embedding scatter at the end of the pass walks the tokens in sorted order: argsort + segment offsets are three
        # small torch kernels, issued now on the second side stream
        order_box, order_ready = {}, None

        def mk_order():
            order_box["o"] = ops.token_order(sv.tokens, V)
        defer_light(mk_order)
        if side2 is not None:
            order_ready = torch.cuda.Event()
            order_ready.record(side2)
        flush_wgrads()
        gE_done = None
        if side is not None:          # embed_bwd below accumulates into gE too: ordered after THIS launch only, not a join
            gE_done = torch.cuda.Event()
            gE_done.record(side)
        dy = ops.gemm_nt(dlogits, sh["Et"], drop_p=p, drop_seed=ss(2))           # [TB, D] (through the final dropout)
        gu, gvb = gv("r_w_bias", (HDt,)), gv("r_r_bias", (HDt,))
        if pad:                       # the kernels accumulate [H, DHp] rows; cropped into the gradients at the end
            gu_t, gvb_t = gu, gvb
            gu, gvb = torch.zeros(HD, device=dev, dtype=F32), torch.zeros(HD, device=dev, dtype=F32)
        u_k, vb_k = self._uv()
        side_mark = [None]

        def defer_last(fn):          # the last layer's dRd chain: second side stream, but AFTER everything the first side
            if side2 is None:        # stream was given for the layers above (they all add into r_w_bias / r_r_bias)
                return fn()
            ev = torch.cuda.Event()
            ev.record(main)
            with torch.cuda.stream(side2):
                side2.wait_event(ev)
                side2.wait_event(side_mark[0])
                fn()

        dy_add = None
        for i in range(L - 1, -1, -1):
            pre = f"layers.{i}."
            lay = self.layers[i]
            s0 = 16 + 4 * i
            if i == 0 and side is not None:
                side_mark[0] = torch.cuda.Event()
                side_mark[0].record(side)
            dz2m = torch.empty(TB, D, device=dev, dtype=BF16) if p > 0 else None
            dz2, part = ops.layernorm_bwd(dy, sv.z2[i], sv.mu2[i], sv.rs2[i], lay.pos_ff.layer_norm.weight,
                                          dz_masked=dz2m, drop_p=p, drop_seed=ss(s0 + 3), add=dy_add)
            dy_add = None
            if dz2m is None:
                dz2m = dz2
            keep.append(part)
            light(lambda part=part, pre=pre: ops.layernorm_bwd_reduce(
                part, gv(pre + "pos_ff.layer_norm.weight", (Dt,)), gv(pre + "pos_ff.layer_norm.bias", (Dt,)),
                gv(pre + "pos_ff.CoreNet.3.bias", (Dt,)), group=cgrp))
            wgrad(dz2m, sv.hid[i], gv(pre + "pos_ff.CoreNet.3.weight", (Dt, DIt)), crop=spec("w2"))
            hb = sv.hbits[i]
            if sv.zff[i] is not None:          # GELU: dz = dhid_raw * keep/(1-p) * gelu'(z)
                dhid = ops.gelu_bwd(ops.gemm_nt(dz2m, sh[f"w2_t{i}"]), sv.zff[i], drop_p=p, drop_seed=ss(s0 + 2))
            elif hb is not None and ops.signbits_words(TB, DI, D, dz2m.stride(0), sh[f"w2_t{i}"].stride(0), DI) == hb.numel():
                dhid = ops.gemm_nt(dz2m, sh[f"w2_t{i}"], relu_bits=hb, mask_scale=inv_keep)
            else:
                dhid = ops.gemm_nt(dz2m, sh[f"w2_t{i}"], relu_mask=sv.hid[i], mask_scale=inv_keep)
            # (the bias gradient = column sums of dhid comes out of the weight-gradient launch: no pass of its own)
            wgrad(dhid, sv.a[i], gv(pre + "pos_ff.CoreNet.0.weight", (DIt, Dt)), crop=spec("w1"),
                  colsum=(gv(pre + "pos_ff.CoreNet.0.bias", (DIt,)), DIt))
            # (model.resid_in_ln_bwd, A/B: the residual branch's gradient as a second addend of the LayerNorm backward that
            #  consumes the sum, instead of an auxiliary operand of the dX GEMM -- which then takes the pipelined epilogue)
            rln = bool(getattr(self, "resid_in_ln_bwd", False))
            da = ops.gemm_nt(dhid, sh[f"w1_t{i}"]) if rln else ops.gemm_nt(dhid, sh[f"w1_t{i}"], resid=dz2)
            dz1m = torch.empty(TB, D, device=dev, dtype=BF16) if p > 0 else None
            dz1, part = ops.layernorm_bwd(da, sv.z1[i], sv.mu1[i], sv.rs1[i], lay.dec_attn.layer_norm.weight,
                                          dz_masked=dz1m, drop_p=p, drop_seed=ss(s0 + 1), add=dz2 if rln else None)
            if dz1m is None:
                dz1m = dz1
            keep.append(part)
            light(lambda part=part, pre=pre: ops.layernorm_bwd_reduce(
                part, gv(pre + "dec_attn.layer_norm.weight", (Dt,)), gv(pre + "dec_attn.layer_norm.bias", (Dt,)), group=cgrp))
            wgrad(dz1m, sv.vec[i], gv(pre + "dec_attn.o_net.weight", (Dt, HDt)), crop=spec("o"))
            last = i == 0
            # Under an overlapped gradient exchange (no column-sum group) the LAST layer's bias chain runs on the second
            # side stream while its qkv weight gradient -- whose slabs would carry the colsum(dq) term of d r_r_bias --
            # is reduced on the first one: two unordered read-modify-writes of one vector.  There the term stays in the
            # bias chain (one stream, stream order), as for every layer before the weight-gradient launch carried it.
            hook_last = last and cgrp is None and side is not None
            if last:
                # the last layer of the pass: nothing follows its weight gradients on the main stream but one GEMM and the
                # embedding scatter, so w2 | w1 | o leave now (beside this layer's attention backward) and only qkv remains
                # for the end; its dRd chain goes to the second side stream, beside that last grouped launch
                flush_wgrads()
            dvec = ops.gemm_nt(dz1m, sh[f"o_t{i}"])
            qkv = sv.qkv[i]
            # (the q third of the memory rows is never written and never read: memory rows have no query, their weight
            #  gradient and dX only take the k|v thirds -- no zero fill)
            dqkv = torch.empty(K * B, 3 * HD, device=dev, dtype=BF16)
            drd = torch.empty(K, HD, device=dev, dtype=F32)
            if scr_free[i & 1] is not None:
                main.wait_event(scr_free[i & 1])          # (layer i+2's dRd GEMM has released this scratch buffer)
            ops.relattn_bwd(qkv[M * B:, :HD], qkv[:, HD:2 * HD], qkv[:, 2 * HD:], sv.rd[i], u_k,
                            vb_k, sv.reset, T, M, B, H, DH, sv.same_length, sv.mem_len, sv.vec[i], dvec,
                            sv.lse[i], sv.qs[i], dqkv[M * B:, :HD], dqkv[:, HD:2 * HD], dqkv[:, 2 * HD:], drd, gu, gvb,
                            drop_p=patt, drop_seed=ss(s0), scale=self.attn_scale,
                            scratch=scr[i & 1], defer=(defer_last if last else defer) if side is not None else None,
                            colsum_group=cgrp, dq_colsum=hook_last)
            gWr = gv(pre + "dec_attn.r_net.weight", (HDt, Dt))
            if side is None:
                self._tn_acc(ops.cast_bf16(drd), sv.pd, gWr, crop=spec("r"))
            else:
                keep.append(drd)
                (defer_last if last else defer)(lambda drd=drd, gWr=gWr, ws="slabs2" if last else "slabs": self._tn_acc(
                    ops.cast_bf16(drd), sv.pd, gWr, crop=spec("r"), ws=ws))
                def mark_free(i=i):          # (after the deferred work above: it is enqueued at the flush below)
                    scr_free[i & 1] = torch.cuda.Event()
                    scr_free[i & 1].record(side)
                if last:
                    mark_free()
                else:
                    defer(mark_free)
            gW = gv(pre + "dec_attn.qkv_net.weight", (3 * HDt, Dt))
            # (d r_r_bias's colsum(dq) term = the first H*DH column sums of dqkv: from the weight-gradient launch)
            wgrad(dqkv[M * B:], sv.h[i], gW, crop=spec("qkv"), colsum=None if hook_last else (gvb, HD))
            if M > 0:
                wgrad(dqkv[:M * B, HD:], sv.cat[i], gW[HDt:], crop=spec("kv"))
            flush_wgrads(wide=last and not getattr(self, "narrow_last_wgrad", False))          # (attribute: A/B runs only)
            if rln and not last:          # (the layer below's LayerNorm backward adds dz1; the embedding scatter needs the sum)
                dy = ops.gemm_nt(dqkv[M * B:], sh[f"qkv_t{i}"])
                dy_add = dz1
            else:
                dy = ops.gemm_nt(dqkv[M * B:], sh[f"qkv_t{i}"], resid=dz1)
            hook = getattr(self, "grad_ready_hook", None)
            if hook is not None and direct:
                # every gradient of layer i is final once what has been enqueued so far on BOTH streams has run: its
                # slice of the flat buffer may be exchanged.  The main stream is NOT joined with the side stream
                # here (that would serialise exactly the work the side stream hides); a hook that can order itself
                # after events (`wants_events`) gets one per stream, others get the old blocking join.
                lo = gname[pre + "dec_attn.qkv_net.weight"]
                hi = gname[f"layers.{i + 1}.dec_attn.qkv_net.weight"] if i + 1 < L else gname["crit.out_layers.0.bias"]
                if getattr(hook, "wants_events", False):
                    evs = [torch.cuda.Event()]
                    evs[0].record(main)
                    if side is not None:
                        evs.append(torch.cuda.Event())
                        evs[1].record(side)
                        evs.append(torch.cuda.Event())
                        evs[2].record(side2)
                    hook(G, lo, hi, evs)
                else:
                    join()
                    hook(G, lo, hi)
        if gE_done is not None:
            main.wait_event(gE_done)          # (gE also received the output-layer weight gradient from the side stream)
        if order_ready is not None:
            main.wait_event(order_ready)
            for t_ in order_box["o"]:
                t_.record_stream(main)          # (allocated on the second side stream, read here)
        ops.embed_bwd(sv.tokens, dy, gE, accumulate=True, drop_p=p, drop_seed=ss(0), order=order_box["o"])
        if cgrp is not None:
            if side is not None:      # the grouped final column sums: after every slab pass on either side stream
                with torch.cuda.stream(side2):
                    side2.wait_stream(side)
                    side2.wait_stream(main)          # (sources that joined the group without a hand-over of their own)
                    cgrp.flush()
            else:
                cgrp.flush()
        flush_wgrads()
        join()
        if pad:
            gu_t.view(H, DHt).add_(gu.view(H, DH)[:, :DHt])
            gvb_t.view(H, DHt).add_(gvb.view(H, DH)[:, :DHt])
        if direct:
            return tuple(None for _ in params)
        return tuple(G[off:off + p.numel()].view(p.shape) for p, off in zip(params, fl["offs"]))

    def _tn_group_acc(self, items, budget=None):
        """items = [(dY, X, gW, rows, crop), ...] with a common token count: gW (+)= dY^T X for each, one grouped
        launch + one reduce per problem; falls back to _tn_acc when the grouped kernel does not take the shapes."""
        fl = self._flat
        arr, Mtok, offs, total, cs_offs = ops.tn_group([(it[0], it[1]) for it in items], colsum=[it[5] is not None for it in items])
        ns = ops.tn_group_slices(arr, Mtok, budget)          # (<= 8 problems; the reductions below go in launches of <= 8 destinations)
        if ns <= 0:
            for dY, Xa, gW, rows, crop, cs in items:
                self._tn_acc(dY, Xa, gW, rows=rows, crop=crop)
                if cs is not None:
                    ops.colsum(dY[:, :cs[1]], cs[0])
            return
        need = ns * total
        if fl.get("slabs") is None or fl["slabs"].numel() < need:
            fl["slabs"] = torch.empty(need, device=fl["dev"], dtype=F32)
        slabs = fl["slabs"]
        ops.gemm_tn_grouped(arr, Mtok, slabs, total, ns)
        red = []
        for (dY, Xa, gW, rows, crop, cs), off, cso in zip(items, offs, cs_offs):
            N, Kc = dY.shape[1], Xa.shape[1]
            if cs is not None:
                red.append((cs[0], cso, (1, 1, 1, 1, cs[1], N)))
            if crop is not None:
                rg, rt, rp, cg, ct, cp = crop
                assert rg * rp == N and cg * cp == Kc, (crop, N, Kc)
            else:
                crop = (1, N if rows is None else rows, N, 1, Kc, Kc)
            red.append((gW, off, crop))
        # One reduce launch for the whole group -- EXCEPT that items whose destinations overlap must not share a launch:
        # with mem_len == tgt_len the memory-side k|v gradient (dst = gW[HDt:]) has the token count of the qkv gradient
        # (dst = gW) and lands in this group; two workgroup sets would read-modify-write the same rows with no ordering.
        # Overlapping items go to consecutive launches (stream order = accumulation order).
        batches = []
        for item in red:
            lo = item[0].data_ptr()
            hi = lo + item[0].numel() * item[0].element_size()
            for bt in batches:
                if len(bt) < 8 and all(hi <= l2 or h2 <= lo for l2, h2, _ in bt):
                    bt.append((lo, hi, item))
                    break
            else:
                batches.append([(lo, hi, item)])
        for bt in batches:
            ops.reduce_slabs_group([it for _, _, it in bt], slabs, ns, total, True)

    def _tn_acc(self, dY, Xa, gW, rows=None, crop=None, ws="slabs"):
        """gW[:rows] += dY^T @ Xa (weight gradient).  gW is a contiguous fp32 view of the flat grads.
        crop = (rg, rt, rp, cg, ct, cp): the product has the padded shape [rg*rp, cg*cp]; its [rt, ct] blocks are
        added to gW [rg*rt, cg*ct].  ws: name of the slab workspace -- one per stream that may run this concurrently."""
        N = dY.shape[1]
        Kc = Xa.shape[1]
        M = dY.shape[0]
        ns = ops.tn_slices(M, N, Kc)
        fl = self._flat
        need = ns * N * Kc
        if fl.get(ws) is None or fl[ws].numel() < need:
            fl[ws] = torch.empty(need, device=fl["dev"], dtype=F32)
        ops.gemm_tn_raw(dY, Xa, fl[ws], ns)
        if crop is not None:
            rg, rt, rp, cg, ct, cp = crop
            assert rg * rp == N and cg * cp == Kc, (crop, N, Kc)
            ops.reduce_slabs_crop(gW, fl[ws], crop, ns, N * Kc, True)
            return
        nrows = N if rows is None else rows
        ops.reduce_slabs(gW, fl[ws], nrows * Kc, ns, N * Kc, True, 1.0)
