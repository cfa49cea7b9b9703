#!/usr/bin/env python3
"""Training driver with the reference's command line (train.py:57-70: --data_dir --work_dir --local_rank):

    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 commu-code_amd/train.py \\
        --data_dir <output_npy> --work_dir <dir>            (one process per MI355X, RCCL over xGMI)
    python commu-code_amd/train.py --data_dir <output_npy> --work_dir <dir>            (single GPU)

What the reference's script does at import time and in train() (:113-288, :357-484) is done here in `main`:
run directory agreed on by a broadcast (C2), per-rank seeds / shuffle seeds / LR scaling (Q7, Q9), the packed-stream
batch iterator, the train step (commu_amd.train.Trainer: micro-batches, clip, Adam, inverse-sqrt LambdaLR, ONE
gradient all-reduce per optimiser step overlapped with backward), the logging window (one packed all-reduce, C5),
evaluation every `eval_interval` steps with the same_length / long-memory setting (C6) and checkpoint_last /
checkpoint_best written by rank 0 between barriers (C7).  The reference's hyper-parameters are hard-coded defaults
(config_helper.py:4-49); the extra flags below only exist so that small runs and the benchmark shapes can be driven
from the same entry point.
"""
import argparse
import math
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="ComMU Transformer-XL training on MI355X")
    p.add_argument("--data_dir", type=str, required=True, help="location of the data corpus (output_npy)")
    p.add_argument("--local_rank", type=int, default=int(os.environ.get("LOCAL_RANK", 0)))
    p.add_argument("--work_dir", type=str, required=True, help="Base directory to save the trained model.")
    # -- not in the reference: overrides of its hard-coded defaults
    p.add_argument("--no-merge-chunks", dest="merge_chunks", action="store_false", default=None,
                   help="(not in the reference) run the batch_chunk micro-batches one after the other like the reference; "
                        "default: one forward / backward over all columns with per-micro-batch loss weights when the batch "
                        "has at most 65536 tokens (same loss and gradients)")
    p.add_argument("--graph", action="store_true",
                   help="replay the optimiser step from hipGraphs once its shapes are steady (not in the reference; pays "
                        "when a micro-batch is small enough for the host's launch rate to bound the step)")
    p.add_argument("--max_step", type=int, default=None)
    p.add_argument("--log_interval", type=int, default=None)
    p.add_argument("--eval_interval", type=int, default=None)
    for name in ("num_layers", "num_heads", "units", "inner_size", "tgt_length", "mem_length", "batch_size",
                 "batch_chunk"):
        p.add_argument("--" + name, type=int, default=None)
    return p.parse_args(argv)


def main(argv=None):
    from commu_amd.ddp import GradReducer
    from commu_amd.model.config_helper import get_default_cfg_training
    from commu_amd.model.dataset import ComMUDataset
    from commu_amd.train import Trainer, build_model, evaluate_best_checkpoint, save_checkpoint
    args = parse_args(argv)
    cfg = get_default_cfg_training()
    cfg.defrost()
    for name in ("num_layers", "num_heads", "units", "inner_size"):
        if getattr(args, name) is not None:
            cfg.MODEL[name] = getattr(args, name)
    for name in ("tgt_length", "mem_length", "batch_size", "batch_chunk", "max_step", "log_interval", "eval_interval"):
        if getattr(args, name) is not None:
            cfg.TRAIN[name] = getattr(args, name)
    cfg.freeze()
    torch.cuda.set_device(args.local_rank)
    device = torch.device("cuda", args.local_rank)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", init_method="env://")          # train.py:361 (RCCL on ROCm)
    rank = dist.get_rank() if world > 1 else 0
    # C2: every rank writes into the run directory named after rank 0's clock (train.py:363-370)
    exp_time = torch.tensor(time.time(), dtype=torch.float64, device=device)
    if world > 1:
        dist.broadcast(exp_time, 0)
    work_dir = os.path.join(args.work_dir, time.strftime("%Y%m%d-%H%M%S", time.localtime(float(exp_time))))
    os.makedirs(work_dir, exist_ok=True)
    if rank == 0:
        with open(os.path.join(work_dir, "config.yml"), "w") as f:
            f.write(str(cfg))

    def log(msg):
        if rank == 0:
            print(msg, flush=True)
        with open(os.path.join(work_dir, f"train_rank{rank}.log"), "a") as f:
            f.write(msg + "\n")

    seed = cfg.TRAIN.seed
    np.random.seed(seed)
    torch.manual_seed(seed)
    dataset = ComMUDataset(args.data_dir, cfg)
    num_gpus = world                                                            # train.py:395 (one process per GPU)
    assert cfg.TRAIN.batch_size % num_gpus == 0
    batch_size = cfg.TRAIN.batch_size // num_gpus
    assert batch_size % cfg.TRAIN.batch_chunk == 0
    train_iter = dataset.get_iterator(batch_size, cfg.TRAIN.tgt_length, device, "train", True,
                                      seed=seed + 1000 * rank)                  # Q9
    val_iter = dataset.eval_iterator(cfg.EVALUATE.batch_size, cfg.EVALUATE.tgt_length, device, "valid",
                                     local_rank=rank, world_size=num_gpus)
    test_iter = dataset.eval_iterator(cfg.EVALUATE.batch_size, cfg.EVALUATE.tgt_length, device, "test",
                                      local_rank=rank, world_size=num_gpus)
    assert cfg.MODEL.units % cfg.MODEL.num_heads == 0
    model = build_model(cfg, dataset.vocab, device)
    log(f"#total params = {sum(p.nelement() for p in model.parameters())}")
    reducer = GradReducer() if world > 1 else None
    if reducer is not None:
        reducer.broadcast_params(model)                                         # DDP constructor semantics (C3)
    trainer = Trainer(model, cfg, num_gpus=num_gpus, reducer=reducer, graph=bool(getattr(args, "graph", False)),
                      merge_chunks=getattr(args, "merge_chunks", None), settle_heap=True)
    best_val_nll = float("inf")

    def checkpoint(name, val_nll):                                              # train.py:29-54 (C7)
        if reducer is not None:
            reducer.barrier()
        if rank == 0:
            save_checkpoint(os.path.join(work_dir, name), model, trainer.optimizer, dataset.vocab, trainer.train_step,
                            val_nll, trainer.scheduler)
        if reducer is not None:
            reducer.barrier()

    log("Start training")
    t_log = time.time()
    for data, target, reset_mems, ntok in train_iter():
        trainer.step(data, target, reset_mems, ntok)
        step = trainer.train_step
        if step % cfg.TRAIN.log_interval == 0:
            nll, gnorm, tokens = trainer.log_window()
            elapsed = time.time() - t_log
            log("Train Step {}/{}, lr={:f}, tokens/s={:.1f}, nll={:.4f}, ppl={:.2f}, grad norm={}, ".format(
                step, cfg.TRAIN.max_step, trainer.optimizer.param_groups[0]["lr"], tokens / elapsed, nll,
                math.exp(min(nll, 50.0)), gnorm))
            t_log = time.time()
        if step % cfg.TRAIN.eval_interval == 0:
            t0 = time.time()
            val_tok, val_nll = trainer.evaluate_reduced(val_iter)
            log("Eval step {}, time={}s, val nll={}, val ppl={},".format(step, time.time() - t0, val_nll,
                                                                        math.exp(min(val_nll, 50.0))))
            checkpoint("checkpoint_last.pt", val_nll)
            if val_nll < best_val_nll:
                best_val_nll = val_nll
                checkpoint("checkpoint_best.pt", best_val_nll)
                t0 = time.time()
                test_tok, test_nll = trainer.evaluate_reduced(test_iter)
                log("Test step {}, time={}s, test nll={}, test ppl={}, #evaluated tokens={}".format(
                    step, time.time() - t0, test_nll, math.exp(min(test_nll, 50.0)), test_tok))
            t_log = time.time()
        if step >= cfg.TRAIN.max_step:
            log("-" * 100)
            log("End of training")
            break
    # train.py:486-513: the best checkpoint, reloaded into a fresh same_length model, on the test split
    best = os.path.join(work_dir, "checkpoint_best.pt")
    if reducer is not None:
        reducer.barrier()
    if os.path.exists(best):
        test_nll, _ = evaluate_best_checkpoint(best, cfg, dataset.vocab, device, test_iter, reducer=reducer)
        log("=" * 100)
        log("| End of training | test nll {:5.2f} | test ppl {:9.3f}".format(test_nll, math.exp(min(test_nll, 50.0))))
        log("=" * 100)
    if world > 1:
        dist.destroy_process_group()
    return work_dir


if __name__ == "__main__":
    main()
