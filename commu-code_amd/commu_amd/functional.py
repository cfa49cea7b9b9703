"""Device-side loss reduction used by the training loop (reference train.py:148-149):
`loss[target != pad].mean()` without the boolean-index host sync."""
from __future__ import annotations

import torch

from . import ops


class _MaskedMean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, nll, target, pad_id, scale):
        dev = nll.device
        ws_sum = torch.empty(1, device=dev, dtype=torch.float32)
        ws_cnt = torch.empty(1, device=dev, dtype=torch.int32)
        out = torch.empty(1, device=dev, dtype=torch.float32)
        n = nll.contiguous().view(-1)
        t = target.contiguous().view(-1)
        ops.masked_mean(n, t, int(pad_id), float(scale), ws_sum, ws_cnt, out)
        ctx.t, ctx.cnt, ctx.pad, ctx.scale, ctx.shape = t, ws_cnt, int(pad_id), float(scale), nll.shape
        ctx.mark_non_differentiable(ws_sum)
        return out[0], ws_sum

    @staticmethod
    def backward(ctx, gout, _gsum):
        g = torch.empty(ctx.t.numel(), device=ctx.t.device, dtype=torch.float32)
        ops.loss_grad(ctx.t, ctx.pad, ctx.cnt, ctx.scale, g)
        return (g * gout).view(ctx.shape), None, None, None


def masked_mean(nll, target, pad_id=0, scale=1.0, with_sum=False):
    """scale * mean(nll[target != pad_id])  -- scale = 1 / batch_chunk in the training loop.
    with_sum: also the raw sum of nll over the non-pad targets (device tensor [1]): what the reference's logging window
    accumulates as loss * token count * batch_chunk (train.py:150-154), without the five small kernels of that expression."""
    loss, nll_sum = _MaskedMean.apply(nll, target, pad_id, scale)
    return (loss, nll_sum) if with_sum else loss


class _MaskedMeanGroups(torch.autograd.Function):
    @staticmethod
    def forward(ctx, nll, target, pad_id, groups):
        dev = nll.device
        T, B = nll.shape
        ws_sum = torch.empty(groups, device=dev, dtype=torch.float32)
        ws_cnt = torch.empty(groups, device=dev, dtype=torch.int32)
        out = torch.empty(1, device=dev, dtype=torch.float32)
        sum_all = torch.empty(1, device=dev, dtype=torch.float32)
        n = nll.contiguous().view(-1)
        t = target.contiguous().view(-1)
        ops.masked_mean_groups(n, t, int(pad_id), 1.0 / groups, B, B // groups, ws_sum, ws_cnt, out, sum_all)
        ctx.t, ctx.cnt, ctx.pad, ctx.groups, ctx.shape = t, ws_cnt, int(pad_id), groups, nll.shape
        ctx.mark_non_differentiable(sum_all)
        return out[0], sum_all

    @staticmethod
    def backward(ctx, gout, _gsum):
        T, B = ctx.shape
        g = torch.empty(ctx.t.numel(), device=ctx.t.device, dtype=torch.float32)
        ops.loss_grad_groups(ctx.t, ctx.pad, ctx.cnt, 1.0 / ctx.groups, B, B // ctx.groups, g)
        return (g * gout).view(ctx.shape), None, None, None


def masked_mean_groups(nll, target, pad_id, groups):
    """nll, target: [T, B] whose columns are `groups` micro-batches of B / groups columns.  Returns
    (sum over the micro-batches of mean(nll[target != pad]) / groups, sum of nll over every non-pad target): what the
    reference's loop over `batch_chunk` micro-batches adds up (train.py:136-155), from one pass over all columns."""
    return _MaskedMeanGroups.apply(nll, target, pad_id, groups)
