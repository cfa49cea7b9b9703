"""Training loop of the hot path: the reference's `train()` / `evaluate()` / initialisation
(train.py:74-110,113-169,291-342,441-461) as importable functions (the reference's train.py is a
script with module-level CUDA/NCCL side effects and cannot be imported).

Differences that do not change the arithmetic:
 * the masked mean loss is reduced on the device (no boolean-index host sync, train.py:148);
 * gradients of the `batch_chunk` micro-batches accumulate in the flat gradient buffer and are
   all-reduced ONCE per optimiser step (the reference all-reduces on every micro-batch, quirk Q8;
   the mean of sums is identical);
 * clip + Adam are two kernels on the flat buffers.
"""
from __future__ import annotations

import gc
from typing import List, Optional

import torch
import torch.nn as nn

from .functional import masked_mean, masked_mean_groups
from .model.model import MemTransformerLM
from .optim import FusedAdam, clip_grad_norm_, lr_lambda_factory


def weights_init(m, cfg):
    """train.py:305-342: N(0, base_init) for Linear/Embedding weights and r_*_bias, 0 for biases,
    N(1, base_init) for LayerNorm weights.  Dispatch on class names, as the reference does."""
    std = cfg.INITIALIZER.base_init
    name = m.__class__.__name__
    if name.find("Linear") != -1:
        if getattr(m, "weight", None) is not None:
            nn.init.normal_(m.weight, 0.0, std)
        if getattr(m, "bias", None) is not None:
            nn.init.constant_(m.bias, 0.0)
    elif name.find("AdaptiveEmbedding") != -1:
        pass                                    # emb_projs is empty (d_proj == d_embed)
    elif name.find("Embedding") != -1:
        if hasattr(m, "weight"):
            nn.init.normal_(m.weight, 0.0, std)
    elif name.find("LayerNorm") != -1:
        if hasattr(m, "weight"):
            nn.init.normal_(m.weight, 1.0, std)
        if getattr(m, "bias", None) is not None:
            nn.init.constant_(m.bias, 0.0)
    elif name.find("TransformerLM") != -1:
        if hasattr(m, "r_w_bias"):
            nn.init.normal_(m.r_w_bias, 0.0, std)
        if hasattr(m, "r_r_bias"):
            nn.init.normal_(m.r_r_bias, 0.0, std)


def build_model(cfg, vocab, device, seed=None):
    """train.py:382-438: seed, construct, weights_init, move to the device."""
    if seed is not None:
        torch.manual_seed(seed)
    model = MemTransformerLM(cfg, vocab)
    model.apply(lambda m: weights_init(m, cfg))
    model.word_emb.apply(lambda m: weights_init(m, cfg))
    return model.to(device)


class Trainer:
    """One rank of the data-parallel training job (train.py:441-473 + train() :113-169)."""

    MERGE_MAX_ROWS = 65536          # tokens (T x B) up to which the micro-batches of a step are run as ONE pass

    def __init__(self, model, cfg, num_gpus=1, reducer=None, pad_id=0, graph=False, merge_chunks=None, settle_heap=False):
        """merge_chunks: the reference splits a batch into `batch_chunk` micro-batches to fit its GPU's memory
        (train.py:113-155); each contributes mean(loss over ITS non-pad targets) / batch_chunk.  With 288 GB of HBM the
        columns of all micro-batches go through ONE forward / backward whose loss weights every token by
        1 / (batch_chunk x non-pad count of its micro-batch) -- the same loss and gradients (tested), 1.5x the throughput
        at the released default config (4 micro-batches of 64 columns x 128 tokens).  None: whenever the batch has at most
        MERGE_MAX_ROWS tokens; False: the reference's loop.
        graph=True: once the step's shapes are steady (XL memory at its full length) the device work of a step is
        captured in two hipGraphs -- [all micro-batches forward + backward] and [clip + Adam + weight shadows] -- and
        replayed; the gradient exchange of a multi-GPU job runs between the two replays.  See _graph_step.
        settle_heap (OPT-IN: a process-wide side effect -- the application's own objects alive at that moment are frozen
        too and cyclic garbage among them is never reclaimed; train.py and bench.py of this package switch it on): after
        the first step (model, optimiser state, module tables all exist) the Python heap is collected
        ONCE and moved to the collector's permanent generation (gc.freeze): a full collection of a process with torch
        loaded walks ~1e6 objects, 70-80 ms on the MI355X host -- fourteen steps' worth at 8 sequences per GPU -- and the
        step's own short-lived containers trigger one every few dozen steps otherwise (profiles/r04_b8_gc.txt)."""
        self._one = None
        self.settle_heap = bool(settle_heap)
        self._heap_settled = False
        self.model, self.cfg, self.num_gpus, self.reducer, self.pad_id = model, cfg, num_gpus, reducer, pad_id
        self.graph_mode = bool(graph)
        self.merge_chunks = merge_chunks
        self.groups = None                # micro-batches folded into one pass (decided at the first step; 1: none)
        self._graphs = None
        self._graph_key = None
        self.graph_failed = None          # reason the capture was given up (then the step stays eager, in this process)
        # dropout salt of a replayed step (the seed arguments are frozen in the graph): torch's CPU generator, no device sync
        self.salt_source = lambda: int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
        local_lr = cfg.TRAIN.lr / num_gpus                                  # train.py:441
        self.optimizer = FusedAdam(model, lr=local_lr, weight_decay=cfg.TRAIN.weight_decay)
        self.scheduler = torch.optim.lr_scheduler.LambdaLR(
            self.optimizer, lr_lambda=lr_lambda_factory(cfg.TRAIN.warmup_step, cfg.TRAIN.lr, cfg.TRAIN.lr_min))
        self.train_step = 0
        self.mems: List[Optional[torch.Tensor]] = [None for _ in range(cfg.TRAIN.batch_chunk)]
        dev = next(model.parameters()).device
        self.log_train_loss = torch.zeros((), device=dev)
        self.log_grad_norm = torch.zeros((), device=dev)
        self.log_token_num = 0
        self._log_step0 = 0

    def step(self, data, target, reset_mems, batch_token_num):
        """train.py:131-169 for one batch; returns the summed micro-batch loss (device tensor)."""
        if self.graph_mode and self.graph_failed is None and self._graph_ready(data, reset_mems):
            return self._graph_step(data, target, reset_mems, batch_token_num)
        cfg, model = self.cfg, self.model
        groups = self._decide_groups(data)
        chunk = cfg.TRAIN.batch_chunk // groups          # passes actually run: 1 when the micro-batches are merged
        model.temperature = 1.0
        model.zero_grad()
        data_chunks = torch.chunk(data, chunk, 1)
        target_chunks = torch.chunk(target, chunk, 1)
        reset_chunks = torch.chunk(reset_mems, chunk, 0)
        total = None
        overlap = self.reducer is not None and getattr(model, "grad_mode", "") == "direct"
        for i in range(chunk):
            d, t, r = data_chunks[i].contiguous(), target_chunks[i].contiguous(), reset_chunks[i].contiguous()
            loss, new_mems = model(d, t, r, self.mems[i])
            if self._graphs is not None and self._graph_state["mems"][i] is not None and new_mems is not None:
                if new_mems.shape == self._graph_state["mems"][i].shape:
                    self._graph_state["mems"][i].copy_(new_mems)  # an eager step between replays: the graph's buffers stay current
                    new_mems = self._graph_state["mems"][i]
                else:                                             # another memory shape: the captured step is stale
                    self._graphs, self._graph_state, self._graph_key = None, None, None
            self.mems[i] = new_mems
            if groups > 1:
                loss, nll_sum = masked_mean_groups(loss, t, self.pad_id, groups)
            else:
                loss, nll_sum = masked_mean(loss, t, self.pad_id, 1.0 / chunk, with_sum=True)
            if overlap and i == chunk - 1:
                # gradients are complete once the LAST micro-batch's backward has passed a layer: exchange that
                # layer's slice while the layers below are still being differentiated
                self.reducer.begin()
                model.grad_ready_hook = self.reducer.range_ready
            try:
                # (a cached scalar 1 as the seed gradient: the engine's own ones_like is a fill launch per micro-batch)
                if self._one is None or self._one.device != loss.device:
                    self._one = torch.ones((), device=loss.device, dtype=loss.dtype)
                loss.backward(self._one)
            finally:
                model.grad_ready_hook = None
            total = loss.detach() if total is None else total + loss.detach()
            # train.py:150-154: the logging window accumulates the SUM of the micro-batch's token NLLs
            # (mean / chunk * count * chunk), kept on the device -- the reference syncs here with .item()
            # (= loss * non-pad count * chunk; the masked-mean kernel has that sum already)
            self.log_train_loss += nll_sum[0]
        if self.reducer is not None:
            if overlap:
                self.reducer.finish(model._ensure_flat()["g"])
            else:
                self.reducer.allreduce_mean(model)
        grad_norm = clip_grad_norm_(model, cfg.TRAIN.clip, self.optimizer)
        self.optimizer.step()
        self.optimizer.zero_grad()
        self.train_step += 1
        self.scheduler.step()
        self.log_grad_norm += grad_norm
        self.log_token_num += int(batch_token_num)
        if self.settle_heap and not self._heap_settled:
            self._settle_heap()
        return total

    def _settle_heap(self):
        gc.collect()
        gc.freeze()
        self._heap_settled = True

    def _decide_groups(self, data):
        """Fold the step's micro-batches into one pass?  Decided once, at the first step (the XL memories are kept per
        pass: self.mems has one entry per pass, covering all of that pass's columns)."""
        if self.groups is None:
            chunk = self.cfg.TRAIN.batch_chunk
            want = self.merge_chunks
            if want is None:
                want = data.shape[0] * data.shape[1] <= self.MERGE_MAX_ROWS
            ok = chunk > 1 and chunk <= 16 and data.shape[1] % chunk == 0
            self.groups = chunk if (want and ok) else 1
            if self.groups > 1:
                self.mems = [None]
        return self.groups

    # ------------------------------------------------------------------ hipGraph step
    def _graph_ready(self, data, reset_mems):
        """Capture / replay only at steady shapes: every micro-batch's XL memory at its full length (or no memory), the
        optimiser state allocated (one eager step done), and the batch shape of the captured graph."""
        cfg, model = self.cfg, self.model
        if self.train_step < 2:
            return False
        if cfg.TRAIN.mem_length > 0:
            for m in self.mems:
                if m is None or m.shape[1] != cfg.TRAIN.mem_length:
                    return False
        key = (tuple(data.shape), data.device, cfg.TRAIN.batch_chunk, bool(model.training), float(model.drop.p))
        if self._graphs is not None and key != self._graph_key:
            return False                                      # another shape: eager (the captured graph stays valid)
        return True

    def _device_forward_backward(self, data, target, reset_mems, mems_in):
        """The device work of train.py:133-155 without the autograd engine: forward schedule, masked mean, its
        gradient, backward schedule -- all launches on the current stream (and the model's side streams, joined)."""
        from . import ops
        cfg, model = self.cfg, self.model
        groups = self._decide_groups(data)
        chunk = cfg.TRAIN.batch_chunk // groups
        model.zero_grad()
        data_chunks = torch.chunk(data, chunk, 1)
        target_chunks = torch.chunk(target, chunk, 1)
        reset_chunks = torch.chunk(reset_mems, chunk, 0)
        total, nll_sums, mems_out = None, [], []
        dev = data.device
        for i in range(chunk):
            d, t, r = data_chunks[i].contiguous(), target_chunks[i].contiguous(), reset_chunks[i].contiguous()
            m_in = mems_in[i]
            if m_in is None:
                m_in = model.init_mems(model.n_layer)
            nll, new_mems, sv = model._run_forward(d, t, r, m_in, need_grad=True)
            out = torch.empty(1, device=dev, dtype=torch.float32)
            tt = t.view(-1)
            g = torch.empty(tt.numel(), device=dev, dtype=torch.float32)
            if groups > 1:
                Bm = t.shape[1]
                ws_grp = torch.empty(groups, device=dev, dtype=torch.float32)
                ws_cnt = torch.empty(groups, device=dev, dtype=torch.int32)
                ws_sum = torch.empty(1, device=dev, dtype=torch.float32)
                ops.masked_mean_groups(nll.view(-1), tt, int(self.pad_id), 1.0 / groups, Bm, Bm // groups, ws_grp, ws_cnt,
                                       out, ws_sum)
                ops.loss_grad_groups(tt, int(self.pad_id), ws_cnt, 1.0 / groups, Bm, Bm // groups, g)
            else:
                ws_sum = torch.empty(1, device=dev, dtype=torch.float32)
                ws_cnt = torch.empty(1, device=dev, dtype=torch.int32)
                ops.masked_mean(nll.view(-1), tt, int(self.pad_id), 1.0 / chunk, ws_sum, ws_cnt, out)
                ops.loss_grad(tt, int(self.pad_id), ws_cnt, 1.0 / chunk, g)
            model._run_backward(sv, g.view(nll.shape))
            total = out[0] if total is None else total + out[0]
            nll_sums.append(ws_sum)
            mems_out.append(new_mems if mems_in[i] is not None or cfg.TRAIN.mem_length > 0 else None)
        return total, nll_sums, mems_out

    def _capture(self, data, target, reset_mems):
        from . import ops
        cfg, model = self.cfg, self.model
        dev = data.device
        fl = model._ensure_flat()
        st = {"data": data.clone(), "target": target.clone(), "reset": reset_mems.clone(),
              "salt": torch.zeros(1, device=dev, dtype=torch.int32),
              "scal": torch.zeros(4, device=dev, dtype=torch.float32),
              "mems": [None if m is None else m.detach().clone() for m in self.mems]}
        cur = torch.cuda.current_stream(dev)
        if fl.get("shadow_ready") is not None:                # recorded by the last eager optimiser step: join it here,
            cur.wait_event(fl["shadow_ready"])                # a captured stream must not wait for an outside event
            fl["shadow_ready"] = None
        torch.cuda.synchronize(dev)
        g_fb, g_opt = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        pool = torch.cuda.graph_pool_handle()
        # (thread_local: the batch producer's worker thread keeps issuing copies on its own stream meanwhile)
        with torch.cuda.graph(g_fb, pool=pool, capture_error_mode="thread_local"):
            ops.set_seed_salt(st["salt"])
            total, nll_sums, mems_out = self._device_forward_backward(st["data"], st["target"], st["reset"], st["mems"])
            for dst, src in zip(st["mems"], mems_out):
                if dst is not None:
                    dst.copy_(src)                            # next replay's memory = this replay's hidden states
            for s_ in nll_sums:
                self.log_train_loss += s_[0]
            ops.set_seed_salt(None)
        self.optimizer.dev_scalars = st["scal"]
        try:
            with torch.cuda.graph(g_opt, pool=pool, capture_error_mode="thread_local"):
                gn = clip_grad_norm_(model, cfg.TRAIN.clip, self.optimizer)
                self.optimizer.step()
                sr = model._flat.get("shadow_ready")
                if sr is not None:                            # the transposed shadows are built on a side stream: join
                    torch.cuda.current_stream(dev).wait_event(sr)
                    model._flat["shadow_ready"] = None
                self.log_grad_norm += gn
        finally:
            self.optimizer.dev_scalars = None
        st["total"] = total
        self._graphs, self._graph_state = (g_fb, g_opt), st
        self._graph_key = (tuple(data.shape), data.device, cfg.TRAIN.batch_chunk, bool(model.training), float(model.drop.p))
        # the capture itself does not execute anything: the caller replays

    def _graph_step(self, data, target, reset_mems, batch_token_num):
        """One optimiser step = copy the batch into the graph's input buffers, write the step's dropout salt and the
        optimiser scalars (lr of the LambdaLR schedule, Adam bias corrections) to device memory, replay.  Host work per
        step: five small launches + two graph launches, whatever the depth of the model."""
        from . import ops
        cfg, model = self.cfg, self.model
        if self._graphs is None:
            err = None
            try:
                self._capture(data, target, reset_mems)
            except Exception as exc:                          # capture refused: stay eager (same process, no re-exec)
                err = f"{type(exc).__name__}: {exc}"
            # the decision is COLLECTIVE: a rank that fell back alone would run the overlapped bucket exchange of the
            # eager step against the other ranks' single post-replay exchange (different collectives: a hang)
            if self.reducer is not None:
                ok_all = self.reducer.all_ok(err is None, device=data.device)
                if not ok_all and err is None:
                    err = "graph capture failed on another rank"
            if err is not None:
                self.graph_failed = err
                self._graphs, self._graph_state = None, None
                torch.cuda.synchronize()
                return self.step(data, target, reset_mems, batch_token_num)
            # hand the live memories to the graph: from now on they live in its static buffers (the list is the
            # trainer's own: an eager step may rebind its entries without touching the graph's)
            self.mems = list(self._graph_state["mems"])
            # (no second gc.freeze after a capture: settling is done once per trainer, after its first step)
        st = self._graph_state
        fl = model._flat
        if fl.get("shadow_ready") is not None:                # an EAGER step ran since the last replay: its transposed
            torch.cuda.current_stream(data.device).wait_event(fl["shadow_ready"])      # shadows must be complete
            fl["shadow_ready"] = None
        st["data"].copy_(data)
        st["target"].copy_(target)
        st["reset"].copy_(reset_mems)
        st["salt"].fill_(int(self.salt_source()))
        opt = self.optimizer
        grp = opt.param_groups[0]
        opt.step_count += 1
        bc1, bc2 = ops.adam_bias_corrections(grp["betas"][0], grp["betas"][1], opt.step_count)
        st["scal"].copy_(torch.tensor([float(grp["lr"]), bc1, bc2, 0.0], dtype=torch.float32), non_blocking=True)
        self._graphs[0].replay()
        if self.reducer is not None:
            self.reducer.allreduce_mean(model)
        self._graphs[1].replay()
        self.train_step += 1
        self.scheduler.step()
        self.log_token_num += int(batch_token_num)
        return st["total"].clone()          # (the graph's own output buffer is overwritten by the next replay)

    def _sum_over_ranks(self, values):
        """One packed all-reduce for a set of scalars (the reference issues one collective per scalar)."""
        dev = next(self.model.parameters()).device
        if self.reducer is not None:
            return self.reducer.sum_scalars(values, device=dev)
        return [float(torch.as_tensor(v)) for v in values]

    def log_window(self):
        """train.py:171-197: close the logging window -- ONE all-reduce of (NLL sum, grad-norm sum, token count)
        instead of three; returns (nll per token, mean grad norm, tokens of all ranks) and resets the window."""
        loss_sum, gnorm_sum, tokens = self._sum_over_ranks([self.log_train_loss, self.log_grad_norm, self.log_token_num])
        steps = max(self.train_step - self._log_step0, 1)
        self.log_train_loss.zero_()
        self.log_grad_norm.zero_()
        self.log_token_num = 0
        self._log_step0 = self.train_step
        nll = loss_sum / max(tokens, 1.0)
        return nll, gnorm_sum / (steps * self.num_gpus), tokens

    def evaluate_reduced(self, eval_iter):
        """evaluate() + the token-count / NLL all-reduce of train.py:209-210,264-265 (one packed collective);
        returns (tokens of all ranks, NLL per token)."""
        tok, nll = self.evaluate(eval_iter)
        tok_all, nll_all = self._sum_over_ranks([tok, nll / 10000.0])
        return int(tok_all), nll_all / (max(tok_all, 1.0) / 10000.0)

    def evaluate(self, eval_iter):
        """train.py:74-110: same_length evaluation with the longer EVALUATE memory."""
        return evaluate(self.model, self.cfg, eval_iter, self.pad_id)


@torch.no_grad()
def evaluate(model, cfg, eval_iter, pad_id=0):
    """train.py:74-110 for any model: eval mode, EVALUATE lengths, same_length on; the TRAIN settings restored after.
    Returns (non-pad target tokens, summed NLL) of this rank's share of the split."""
    model.eval()
    model.reset_length(tgt_len=cfg.EVALUATE.tgt_length, mem_len=cfg.EVALUATE.mem_length)
    model.same_length = True
    total_tok, total_nll = 0, 0.0
    mems = None
    for data, target, all_reset, ntok in eval_iter():
        if all_reset:
            mems = None
        loss, mems = model(data, target, None, mems)
        total_nll += ntok * float(masked_mean(loss, target, pad_id, 1.0))
        total_tok += ntok
    model.reset_length(cfg.TRAIN.tgt_length, cfg.TRAIN.mem_length)
    model.same_length = cfg.MODEL.same_length
    model.train()
    return total_tok, total_nll


def evaluate_best_checkpoint(path, cfg, vocab, device, eval_iter, reducer=None, pad_id=0):
    """train.py:486-513, the script's last block: a FRESH model with `MODEL.same_length = True`, the weights of
    `checkpoint_best.pt`, the test split through evaluate(), token count and NLL summed over the ranks (one packed
    collective instead of the reference's two); returns (test nll per token, tokens of all ranks)."""
    cfg = cfg.clone()
    cfg.defrost()
    cfg.MODEL.same_length = True
    cfg.freeze()
    model = MemTransformerLM(cfg, vocab)
    model.load_state_dict(read_checkpoint(path)["model"])
    model = model.to(device)
    tok, nll = evaluate(model, cfg, eval_iter, pad_id)
    if reducer is not None:
        tok, nll = reducer.sum_scalars([tok, nll / 10000.0], device=device)
    else:
        tok, nll = float(tok), nll / 10000.0
    return nll / (max(tok, 1.0) / 10000.0), int(tok)


def save_checkpoint(path, model, optimizer, vocab, train_step, best_val_loss, scheduler):
    """train.py:29-54: same dictionary layout (model = unwrapped state_dict, optimizer in torch.optim.Adam's
    layout, amp = None) -- loadable by the reference and vice versa."""
    torch.save({"model": {k: v.detach().float().cpu() for k, v in model.state_dict().items()},
                "optimizer": optimizer.state_dict() if optimizer is not None else None,
                "train_step": train_step, "scheduler": scheduler.state_dict(),
                "best_val_loss": best_val_loss, "vocab": vocab, "amp": None}, path)


class _ReferencePickle:
    """pickle-module stand-in for torch.load: a checkpoint written by the reference's train.py (:39-48) pickles its
    `vocab` object as `commu.model.dataset.BaseVocab`; that package is not needed to read the file -- class lookups
    under `commu.` fall back to this package's restatements."""
    import pickle as _pickle
    __name__ = "pickle"
    load = staticmethod(_pickle.load)

    class Unpickler(_pickle.Unpickler):
        def find_class(self, module, name):
            try:
                return super().find_class(module, name)
            except (ImportError, AttributeError):
                if module == "commu.model.dataset" and name == "BaseVocab":
                    from .model.dataset import BaseVocab
                    return BaseVocab
                raise


def read_checkpoint(path, map_location="cpu"):
    """The checkpoint dictionary of train.py:39-48 ({model, optimizer, train_step, scheduler, best_val_loss, vocab,
    amp}), whether this package or the reference wrote it."""
    return torch.load(path, map_location=map_location, weights_only=False, pickle_module=_ReferencePickle)


def load_checkpoint(path, model, optimizer=None, scheduler=None):
    """Resume from a checkpoint written by save_checkpoint or by the reference (train.py:493-495 reads
    checkpoint["model"]).  Returns (train_step, best_val_loss)."""
    ck = read_checkpoint(path)
    model.load_state_dict(ck["model"])
    if optimizer is not None and ck.get("optimizer") is not None:
        optimizer.load_state_dict(ck["optimizer"])
    if scheduler is not None and ck.get("scheduler") is not None:
        scheduler.load_state_dict(ck["scheduler"])
    return ck.get("train_step", 0), ck.get("best_val_loss")
