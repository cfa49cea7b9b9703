"""Oracle (test infrastructure, see oracle/__init__.py): fp32 PyTorch restatement of
the reference Transformer-XL forward / loss / memory update, written as explicit
index math over a flat ``{state_dict name: tensor}`` dictionary.

Every function cites the reference lines it follows (paths relative to
/root/reference).  Pinned by tests/test_oracle_golden.py against fixtures the
reference itself produced (tests/golden/make_golden.py).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import torch

Tensor = torch.Tensor
LN_EPS = 1e-5  # torch.nn.LayerNorm default, commu/model/model.py:171,214


@dataclass(frozen=True)
class XLShape:
    """Hyper-parameters read by MemTransformerLM.__init__ (commu/model/model.py:429-444)."""
    n_layer: int
    n_head: int
    d_model: int
    d_inner: int
    n_token: int = 729  # event_tokens.py:329 VOCAB_SIZE

    @property
    def d_head(self) -> int:
        return self.d_model // self.n_head


def param_shapes(s: XLShape) -> List[Tuple[str, Tuple[int, ...]]]:
    """state_dict names/shapes of the trainable tensors (commu/model/model.py:451-492).

    ``crit.out_layers.0.weight`` is tied to the embedding (model.py:480-481) and is
    therefore not listed separately.
    """
    D, H, dh, DI, V = s.d_model, s.n_head, s.d_head, s.d_inner, s.n_token
    out = [("word_emb.emb_layers.0.weight", (V, D)),
           ("r_w_bias", (H, dh)), ("r_r_bias", (H, dh))]
    for i in range(s.n_layer):
        p = f"layers.{i}."
        out += [(p + "dec_attn.qkv_net.weight", (3 * H * dh, D)),
                (p + "dec_attn.o_net.weight", (D, H * dh)),
                (p + "dec_attn.layer_norm.weight", (D,)),
                (p + "dec_attn.layer_norm.bias", (D,)),
                (p + "dec_attn.r_net.weight", (H * dh, D)),
                (p + "pos_ff.CoreNet.0.weight", (DI, D)),
                (p + "pos_ff.CoreNet.0.bias", (DI,)),
                (p + "pos_ff.CoreNet.3.weight", (D, DI)),
                (p + "pos_ff.CoreNet.3.bias", (D,)),
                (p + "pos_ff.layer_norm.weight", (D,)),
                (p + "pos_ff.layer_norm.bias", (D,))]
    out += [("crit.out_layers.0.bias", (V,))]
    return out


def init_params(s: XLShape, seed: int, std: float = 0.01,
                dtype=torch.float32) -> Dict[str, Tensor]:
    """Random parameters with the distribution of train.py:291-342 (weights_init):
    Linear/Embedding weights and r_*_bias ~ N(0, std), biases 0, LayerNorm weight ~ N(1, std).
    (Same distribution, not the same RNG stream as the reference.)"""
    g = torch.Generator().manual_seed(seed)
    p: Dict[str, Tensor] = {}
    for name, shape in param_shapes(s):
        if name.endswith("layer_norm.weight"):
            t = 1.0 + std * torch.randn(shape, generator=g)
        elif name.endswith(".bias") and "r_" not in name.split(".")[-1]:
            t = torch.zeros(shape)
        else:
            t = std * torch.randn(shape, generator=g)
        p[name] = t.to(dtype)
    return p


def sinusoid_table(klen: int, d_model: int, dtype=torch.float32, clamp_len: int = -1) -> Tensor:
    """P[k] = [sin(pos_k f) | cos(pos_k f)], pos_k = klen-1-k, clamped to clamp_len when that is > 0
    (model.py:142-147,578-583)."""
    pos = torch.arange(klen - 1, -1, -1.0, dtype=dtype)
    if clamp_len > 0:
        pos = pos.clamp(max=clamp_len)
    inv_freq = 1.0 / (10000 ** (torch.arange(0.0, d_model, 2.0, dtype=dtype) / d_model))
    ang = pos[:, None] * inv_freq[None, :]
    return torch.cat([ang.sin(), ang.cos()], dim=-1)


def attn_mask(qlen: int, mlen: int, bsz: int, reset: Optional[Tensor],
              same_length: bool, mem_len: int) -> Tensor:
    """Boolean [B, T, K] mask, True = masked (model.py:549-574).

    causal: j > i + mlen.  same_length: additionally j <= i - s with
    s = qlen - (klen - mem_len) if klen > mem_len else qlen.  reset[b]: all j < mlen.
    ``mem_len`` is the model attribute, not the actual mlen."""
    klen = qlen + mlen
    i = torch.arange(qlen)[:, None]
    j = torch.arange(klen)[None, :]
    m = j > i + mlen
    if same_length:
        mask_len = klen - mem_len
        s = qlen - mask_len if mask_len > 0 else qlen
        m = m | (j <= i - s)
    m = m[None].repeat(bsz, 1, 1)
    if reset is not None and mlen > 0:
        m[reset.bool().cpu(), :, :mlen] = True
    return m


def layer_norm(x: Tensor, g: Tensor, b: Tensor) -> Tensor:
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + LN_EPS) * g + b


def rel_attention_scores(q: Tensor, k: Tensor, r: Tensor, u: Tensor, v: Tensor) -> Tensor:
    """S[b,n,i,j] = (q_i+u_n).k_j + (q_i+v_n).r[j+T-1-i]   before scaling (model.py:313-325,251-259).

    q [T,B,H,dh], k [K,B,H,dh], r [K,H,dh].  The reference's _rel_shift holds wrapped
    values where j > i + M; they are always masked, here they are left at BD = 0."""
    T, K = q.shape[0], k.shape[0]
    M = K - T
    AC = torch.einsum("ibnd,jbnd->bnij", q + u, k)
    BDp = torch.einsum("ibnd,mnd->bnim", q + v, r)
    i = torch.arange(T)[:, None]
    j = torch.arange(K)[None, :]
    idx = j + (T - 1) - i                      # rel row of r used by (i, j)
    valid = j <= i + M
    idx = idx.clamp(0, K - 1)
    BD = torch.gather(BDp, 3, idx[None, None].expand(BDp.shape[0], BDp.shape[1], T, K))
    BD = BD * valid[None, None]
    return AC + BD


def _nodrop(site, x):
    return x


def attn_block(p: Dict[str, Tensor], li: int, s: XLShape, h: Tensor, mem: Optional[Tensor],
               pos: Tensor, mask: Tensor, drop=_nodrop) -> Tensor:
    """RelPartialLearnableMultiHeadAttn.forward in eval mode (model.py:280-354):
    fused QKV over [mem; h], R projection, rel-pos scores, masked softmax, PV, o_net, post-LN."""
    pre = f"layers.{li}."
    T, B, D = h.shape
    H, dh = s.n_head, s.d_head
    c = h if mem is None or mem.numel() == 0 else torch.cat([mem, h], 0)
    K = c.shape[0]
    heads = c @ p[pre + "dec_attn.qkv_net.weight"].t()
    q, k, v = heads.split(H * dh, dim=-1)
    q = q[-T:].reshape(T, B, H, dh)
    k = k.reshape(K, B, H, dh)
    v = v.reshape(K, B, H, dh)
    r = (pos @ p[pre + "dec_attn.r_net.weight"].t()).reshape(K, H, dh)
    S = rel_attention_scores(q, k, r, p["r_w_bias"], p["r_r_bias"]) * (1.0 / math.sqrt(dh))
    S = S.masked_fill(mask[:, None], float("-inf"))
    A = drop(("att", li), torch.softmax(S, dim=3))                      # dropatt, model.py:337
    o = torch.einsum("bnij,jbnd->ibnd", A, v).reshape(T, B, H * dh)
    return layer_norm(h + drop(("o", li), o @ p[pre + "dec_attn.o_net.weight"].t()),      # drop, model.py:349
                      p[pre + "dec_attn.layer_norm.weight"], p[pre + "dec_attn.layer_norm.bias"])


def ff_block(p: Dict[str, Tensor], li: int, a: Tensor, drop=_nodrop) -> Tensor:
    """PositionwiseFF.forward (model.py:163-181): Linear-ReLU-Dropout-Linear-Dropout, post-LN."""
    pre = f"layers.{li}."
    z = a @ p[pre + "pos_ff.CoreNet.0.weight"].t() + p[pre + "pos_ff.CoreNet.0.bias"]
    # (parity tests of a reduced-precision build may inject ITS ReLU gates -- `drop.relu_gate(li, z)` -> 0/1 tensor --
    #  so that a pre-activation within rounding of 0 does not flip a whole gradient term; default: the reference's ReLU)
    gate = getattr(drop, "relu_gate", None)
    if getattr(drop, "activation", "relu") == "gelu":      # non-default option of the build (the reference has ReLU only)
        act = torch.nn.functional.gelu(z)
    else:
        act = torch.relu(z) if gate is None else z * gate(li, z)
    f = drop(("hid", li), act)
    f = drop(("out", li), f @ p[pre + "pos_ff.CoreNet.3.weight"].t() + p[pre + "pos_ff.CoreNet.3.bias"])
    return layer_norm(a + f, p[pre + "pos_ff.layer_norm.weight"], p[pre + "pos_ff.layer_norm.bias"])


def decoder_layer(p: Dict[str, Tensor], li: int, s: XLShape, h: Tensor, mem: Optional[Tensor],
                  pos: Tensor, mask: Tensor, drop=_nodrop) -> Tensor:
    """RelPartialLearnableDecoderLayer.forward (model.py:370-377)."""
    return ff_block(p, li, attn_block(p, li, s, h, mem, pos, mask, drop), drop)


def forward_hidden(p: Dict[str, Tensor], s: XLShape, tokens: Tensor, reset: Optional[Tensor],
                   mems: Optional[Tensor], mem_len: int, same_length: bool, drop=_nodrop, clamp_len: int = -1
                   ) -> Tuple[Tensor, Optional[Tensor]]:
    """MemTransformerLM._forward in eval mode (model.py:540-604).

    mems: None (mem_len == 0), or [L+1, M, B, D] (M may be 0: the reference's
    ``init_mems`` empty tensor, model.py:498-505).  Returns (hidden [T,B,D], new_mems)."""
    T, B = tokens.shape
    D = s.d_model
    E = p["word_emb.emb_layers.0.weight"]
    h = drop(("emb", 0), E[tokens] * math.sqrt(D))                      # model.py:585
    M = 0 if mems is None or mems.numel() == 0 else mems.shape[1]
    K = T + M
    mask = attn_mask(T, M, B, reset, same_length, mem_len)
    pos = drop(("pos", 0), sinusoid_table(K, D, h.dtype, clamp_len))    # model.py:581-586
    hids = [h]
    for li in range(s.n_layer):
        mem = None if M == 0 else mems[li]
        h = decoder_layer(p, li, s, h, mem, pos, mask, drop)
        hids.append(h)
    new_mems = None
    if mems is not None:
        # model.py:522-536: keep the last mem_len of (old mems ++ new hids), detached
        stacked = torch.stack(hids).detach()
        cat = stacked if M == 0 else torch.cat([mems, stacked], dim=1)
        end = M + T
        beg = max(0, end - mem_len)
        new_mems = cat[:, beg:end]
    return drop(("final", 0), h), new_mems                              # model.py:601 (mems hold the un-dropped h)


def init_mems(s: XLShape, mem_len: int, dtype=torch.float32) -> Optional[Tensor]:
    """model.py:498-505: empty [L+1, 0] tensor when mem_len > 0, else None."""
    return torch.empty(s.n_layer + 1, 0, dtype=dtype) if mem_len > 0 else None


def logits_from_hidden(p: Dict[str, Tensor], hidden: Tensor) -> Tensor:
    """Tied-weight output layer (model.py:44-51,480-481)."""
    return hidden @ p["word_emb.emb_layers.0.weight"].t() + p["crit.out_layers.0.bias"]


def forward_loss(p, s: XLShape, data: Tensor, target: Tensor, reset, mems, mem_len: int,
                 same_length: bool, drop=_nodrop, clamp_len: int = -1) -> Tuple[Tensor, Optional[Tensor]]:
    """MemTransformerLM.forward (model.py:678-693): per-token NLL [T,B] and new mems.
    `drop(site, x)` (optional) applies a dropout mask at the reference's nn.Dropout sites."""
    if mems is None:
        mems = init_mems(s, mem_len, p["r_w_bias"].dtype)
    hidden, new_mems = forward_hidden(p, s, data, reset, mems, mem_len, same_length, drop, clamp_len)
    logits = logits_from_hidden(p, hidden)
    lse = torch.logsumexp(logits, dim=-1)
    nll = lse - torch.gather(logits, 2, target[..., None]).squeeze(-1)
    return nll, new_mems


def forward_generate(p, s: XLShape, data: Tensor, mems, mem_len: int, same_length: bool = True,
                     clamp_len: int = -1) -> Tuple[Tensor, Optional[Tensor]]:
    """MemTransformerLM.forward_generate (model.py:606-628): logits [T,B,V] and new mems."""
    if mems is None:
        mems = init_mems(s, mem_len, p["r_w_bias"].dtype)
    hidden, new_mems = forward_hidden(p, s, data, None, mems, mem_len, same_length, clamp_len=clamp_len)
    return logits_from_hidden(p, hidden), new_mems


def masked_mean_loss(nll: Tensor, target: Tensor, batch_chunk: int = 1, pad_id: int = 0) -> Tensor:
    """train.py:148-149: mean NLL over non-pad targets, divided by batch_chunk."""
    return nll[target != pad_id].float().mean() / batch_chunk


# ----------------------------------------------------------------------------------------------
# optimiser / schedule restatement (train.py:159-169,441-461)
# ----------------------------------------------------------------------------------------------

def lr_lambda(step: int, warmup_step: int = 100, lr: float = 0.004, lr_min: float = 0.0001) -> float:
    """train.py:448-460."""
    if step == 0 and warmup_step == 0:
        return 1.0
    if step > warmup_step:
        return max((warmup_step ** 0.5) / (step ** 0.5), lr_min / lr)
    return step / warmup_step


def clip_coef(grads: List[Tensor], max_norm: float) -> Tuple[Tensor, Tensor]:
    """torch.nn.utils.clip_grad_norm_ (train.py:159-161): returns (total_norm, coef<=1)."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads)).float()
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    return total, coef


@dataclass
class AdamState:
    step: int
    m: Dict[str, Tensor]
    v: Dict[str, Tensor]


def adam_init(p: Dict[str, Tensor]) -> AdamState:
    return AdamState(0, {k: torch.zeros_like(t) for k, t in p.items()},
                     {k: torch.zeros_like(t) for k, t in p.items()})


def adam_update(p: Dict[str, Tensor], g: Dict[str, Tensor], st: AdamState, lr: float,
                b1: float = 0.9, b2: float = 0.999, eps: float = 1e-8) -> None:
    """torch.optim.Adam, weight_decay 0, amsgrad off (train.py:442-443), in place."""
    st.step += 1
    bc1 = 1.0 - b1 ** st.step
    bc2 = 1.0 - b2 ** st.step
    for k in p:
        st.m[k].mul_(b1).add_(g[k], alpha=1 - b1)
        st.v[k].mul_(b2).addcmul_(g[k], g[k], value=1 - b2)
        denom = st.v[k].sqrt() / math.sqrt(bc2) + eps
        p[k].addcdiv_(st.m[k], denom, value=-lr / bc1)


def train_step(p: Dict[str, Tensor], st: AdamState, s: XLShape, data: Tensor, target: Tensor,
               reset: Tensor, mems: List[Optional[Tensor]], *, batch_chunk: int, mem_len: int,
               same_length: bool, lr_now: float, clip: float
               ) -> Tuple[float, float, List[Optional[Tensor]], Dict[str, Tensor]]:
    """One optimiser step of train.py:133-169 in eval-mode arithmetic (dropout 0):
    chunk the batch columns, accumulate grads of mean-NLL/batch_chunk, clip, Adam.
    Returns (summed chunk losses, grad norm, new mems per chunk, clipped grads)."""
    names = list(p.keys())
    leaves = {k: p[k].detach().clone().requires_grad_(True) for k in names}
    acc = {k: torch.zeros_like(p[k]) for k in names}
    loss_sum = 0.0
    new_mems: List[Optional[Tensor]] = []
    dch = torch.chunk(data, batch_chunk, 1)
    tch = torch.chunk(target, batch_chunk, 1)
    rch = torch.chunk(reset, batch_chunk, 0)
    for c in range(batch_chunk):
        nll, nm = forward_loss(leaves, s, dch[c].contiguous(), tch[c].contiguous(),
                               rch[c].contiguous(), mems[c], mem_len, same_length)
        loss = masked_mean_loss(nll, tch[c], batch_chunk)
        gs = torch.autograd.grad(loss, [leaves[k] for k in names])
        for k, g in zip(names, gs):
            acc[k] += g
        loss_sum += float(loss)
        new_mems.append(nm)
    total, coef = clip_coef(list(acc.values()), clip)
    for k in names:
        acc[k] *= coef
    adam_update(p, acc, st, lr_now)
    return loss_sum, float(total), new_mems, acc
