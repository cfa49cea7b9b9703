"""Optimiser side of the training step (reference train.py:159-169,441-461) on the model's flat
fp32 buffers: global-norm clip + Adam as two kernels, no host synchronisation.

`FusedAdam` is a torch.optim.Optimizer (so LambdaLR drives `param_groups[0]["lr"]` exactly as in
the reference, including lr == 0 on step 0), `clip_grad_norm_` mirrors
torch.nn.utils.clip_grad_norm_ (returns the total norm as a device tensor)."""
from __future__ import annotations

import torch

from . import ops
from ._lib import CommuHipError


def lr_lambda_factory(warmup_step, lr, lr_min):
    """Inverse-sqrt schedule multiplier of the reference (train.py:448-460)."""
    def lr_lambda(step):
        if step == 0 and warmup_step == 0:
            return 1.0
        if step > warmup_step:
            return max((warmup_step ** 0.5) / (step ** 0.5), lr_min / lr)
        return step / warmup_step
    return lr_lambda


def _flat_with_grads(model):
    fl = model._ensure_flat()
    params, offs = fl["params"], fl["offs"]
    base = fl["g"].data_ptr()
    for p, off in zip(params, offs):
        if p.grad is None:
            fl["g"][off:off + p.numel()].zero_()
        elif p.grad.data_ptr() != base + 4 * off:           # gradient produced outside the flat buffer
            fl["g"][off:off + p.numel()].copy_(p.grad.reshape(-1))
    return fl


def clip_grad_norm_(model, max_norm, optimizer=None):
    """Global L2 norm of all gradients; scales them by min(1, max_norm/(norm+1e-6)).
    With a FusedAdam `optimizer` the scaling is folded into its next step() (same arithmetic)."""
    fl = _flat_with_grads(model)
    ops.grad_norm(fl["g"], fl["gpart"], fl["gnorm"])
    if isinstance(optimizer, FusedAdam):
        optimizer._pending_clip = float(max_norm)
    else:
        ops.scale_clip(fl["g"], fl["gnorm"], float(max_norm))
    return fl["gnorm"][0]


class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Adam(lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0) semantics (train.py:442-443)
    as one kernel over the flat parameter buffer; also refreshes the bf16 weight shadows."""

    def __init__(self, model, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        if weight_decay != 0.0:
            raise CommuHipError("weight_decay != 0 is not used by the reference (config_helper.py:33)")
        super().__init__(list(model.parameters()), dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.model = model
        self.step_count = 0
        self._pending_clip = None
        self.dev_scalars = None          # fp32 device tensor {lr, 1 - b1^t, 1 - b2^t}: see step()

    @torch.no_grad()
    def step(self, closure=None):
        fl = _flat_with_grads(self.model)
        if fl["m"] is None:
            fl["m"] = torch.zeros_like(fl["p"])
            fl["v"] = torch.zeros_like(fl["p"])
        grp = self.param_groups[0]
        clip = self._pending_clip
        self._pending_clip = None
        if self.dev_scalars is not None:
            # graph-replayable form: lr and the bias corrections come from device memory (the owner of the graph writes
            # them -- and advances step_count -- before every replay: Trainer._graph_step)
            ops.adam_step_dev(fl["p"], fl["g"], fl["m"], fl["v"], fl["bf16"], self.dev_scalars,
                              gnorm=fl["gnorm"] if clip is not None else None, clip=clip or 0.0,
                              beta1=grp["betas"][0], beta2=grp["betas"][1], eps=grp["eps"])
        else:
            self.step_count += 1
            ops.adam_step(fl["p"], fl["g"], fl["m"], fl["v"], fl["bf16"], float(grp["lr"]), self.step_count,
                          gnorm=fl["gnorm"] if clip is not None else None, clip=clip or 0.0,
                          beta1=grp["betas"][0], beta2=grp["betas"][1], eps=grp["eps"])
        self.model._refresh_shadows(cast=False)
        fl["version"] = sum(p._version for p in fl["params"])

    # ---- checkpointing in torch.optim.Adam's own layout (train.py:39-41 saves optimizer.state_dict()): state is
    # indexed by parameter position, which is model.parameters() order here as in the reference
    def state_dict(self):
        fl = self.model._ensure_flat()
        grp = {k: v for k, v in self.param_groups[0].items() if k != "params"}
        for k, v in (("amsgrad", False), ("maximize", False), ("foreach", None), ("capturable", False),
                     ("differentiable", False), ("fused", None)):
            grp.setdefault(k, v)
        grp["params"] = list(range(len(fl["params"])))
        state = {}
        if fl["m"] is not None:
            for idx, (p, off) in enumerate(zip(fl["params"], fl["offs"])):
                n = p.numel()
                state[idx] = {"step": torch.tensor(float(self.step_count)),
                              "exp_avg": fl["m"][off:off + n].view(p.shape).detach().cpu().clone(),
                              "exp_avg_sq": fl["v"][off:off + n].view(p.shape).detach().cpu().clone()}
        return {"state": state, "param_groups": [grp]}

    @torch.no_grad()
    def load_state_dict(self, sd):
        fl = self.model._ensure_flat()
        groups = sd["param_groups"]
        if len(groups) != 1 or len(groups[0]["params"]) != len(fl["params"]):
            raise CommuHipError("optimizer state does not match this model (one group, %d parameters expected)"
                                % len(fl["params"]))
        for k in ("lr", "betas", "eps", "weight_decay", "initial_lr"):
            if k in groups[0]:
                self.param_groups[0][k] = tuple(groups[0][k]) if k == "betas" else groups[0][k]
        if float(self.param_groups[0].get("weight_decay", 0.0)) != 0.0:
            raise CommuHipError("weight_decay != 0 is not supported")
        state = sd.get("state", {})
        if not state:
            fl["m"], fl["v"], self.step_count = None, None, 0
            return
        fl["m"] = torch.zeros_like(fl["p"])
        fl["v"] = torch.zeros_like(fl["p"])
        steps = set()
        for idx, (p, off) in enumerate(zip(fl["params"], fl["offs"])):
            st = state.get(idx, state.get(str(idx)))
            if st is None:
                raise CommuHipError(f"optimizer state misses parameter {idx}")
            n = p.numel()
            if tuple(st["exp_avg"].shape) != tuple(p.shape):
                raise CommuHipError(f"optimizer state of parameter {idx} has shape {tuple(st['exp_avg'].shape)}")
            fl["m"][off:off + n].copy_(st["exp_avg"].reshape(-1).to(fl["m"].device, torch.float32))
            fl["v"][off:off + n].copy_(st["exp_avg_sq"].reshape(-1).to(fl["v"].device, torch.float32))
            steps.add(int(float(st["step"])))
        if len(steps) != 1:
            raise CommuHipError("per-parameter step counts differ: not an Adam state this kernel can resume")
        self.step_count = steps.pop()
