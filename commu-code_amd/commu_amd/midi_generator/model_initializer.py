"""Checkpoint -> generation-ready model (the reference's ModelInitializeTask,
commu/midi_generator/model_initializer.py:13-56).

`execute()` does what the reference does with a checkpoint file: build `MemTransformerLM` from the training
defaults with `same_length = True` (:36-41), load `checkpoint["model"]` with `strict=False` (:43-47), move to the
device, `eval()`, `reset_length(1, inference_cfg.MODEL.memory_length)` (:48-50).  The file may have been written
by the reference's train.py (:29-54) -- its pickle then names `commu.model.dataset.BaseVocab`, which
`commu_amd.train.read_checkpoint` maps onto this package's class -- or by `commu_amd.train.save_checkpoint`.
The model this returns runs on the GPU only (no CPU fallback).
"""
from __future__ import annotations

from pathlib import Path
from typing import Optional, Tuple

import torch

from ..model.config_helper import get_default_cfg_inference, get_default_cfg_training
from ..model.dataset import BaseVocab
from ..model.model import MemTransformerLM
from ..train import read_checkpoint


class ModelInitializeTask:
    def __init__(self, model_args, map_location: str = "cpu", device: Optional[torch.device] = None,
                 training_cfg=None):
        """model_args: anything with `.checkpoint_dir` (path of the checkpoint FILE, as in generate.py --checkpoint_dir).
        training_cfg: override of the training defaults for checkpoints of another shape (the reference always
        rebuilds the default shape, quirk Q11)."""
        self.model_args = model_args
        self.map_location = map_location
        self.device = device if device is not None else torch.device("cuda")
        self.inference_cfg = self.initialize_inference_config()
        self._training_cfg = training_cfg

    def initialize_inference_config(self):
        cfg = get_default_cfg_inference()
        cfg.freeze()
        return cfg

    def load_checkpoint_fp(self) -> Tuple[Path, Path]:                       # :25-34
        checkpoint_dir = getattr(self.model_args, "checkpoint_dir", None)
        if checkpoint_dir:
            model_fp = Path(checkpoint_dir)
            return model_fp, model_fp.parent / "config.yml"
        # (the reference falls back to inference_cfg.MODEL.model_directory / checkpoint_name here, fields its own
        #  config does not define: config_helper.py:61-80)
        raise ValueError("--checkpoint_dir (path of the checkpoint file) is required")

    def initialize_training_cfg(self):                                        # :36-41
        cfg = self._training_cfg if self._training_cfg is not None else get_default_cfg_training()
        cfg.defrost()
        cfg.MODEL.same_length = True
        cfg.freeze()
        return cfg

    def initialize_model(self, training_cfg, model_fp):                       # :43-51
        model = MemTransformerLM(training_cfg, BaseVocab())
        checkpoint = read_checkpoint(model_fp, self.map_location)
        model.load_state_dict(checkpoint["model"], strict=False)
        model = model.to(self.device)
        model.eval()
        model.reset_length(1, self.inference_cfg.MODEL.memory_length)
        # generate.py --parity (not in the reference): the fp32 arithmetic of the reference instead of bf16 operands
        model.parity_fp32 = bool(getattr(self.model_args, "parity", False))
        return model

    def execute(self):                                                        # :53-56
        model_fp, _ = self.load_checkpoint_fp()          # (config.yml next to it is never read: quirk Q11)
        return self.initialize_model(self.initialize_training_cfg(), model_fp)
