"""Configuration surface of the hot path: the same field names and defaults as the reference's
yacs config (commu/model/config_helper.py:4-80), held in a small attribute-dict (yacs is not a
dependency).  `get_cfg(...)` builds the re-parameterised shapes BASELINE.json names."""
from __future__ import annotations


class CfgNode(dict):
    """Attribute-access dict with yacs-style freeze()/defrost()."""

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        object.__setattr__(self, "_frozen", False)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        if object.__getattribute__(self, "_frozen"):
            raise AttributeError(f"Attempted to set {k} on a frozen CfgNode")
        self[k] = v

    def _set_frozen(self, flag):
        object.__setattr__(self, "_frozen", flag)
        for v in self.values():
            if isinstance(v, CfgNode):
                v._set_frozen(flag)

    def freeze(self):
        self._set_frozen(True)

    def defrost(self):
        self._set_frozen(False)

    def clone(self):
        """Deep copy (yacs CfgNode.clone), defrosted state preserved per node."""
        out = CfgNode()
        for k, v in self.items():
            out[k] = v.clone() if isinstance(v, CfgNode) else v
        object.__setattr__(out, "_frozen", object.__getattribute__(self, "_frozen"))
        return out

    def __str__(self):
        def fmt(node, ind):
            out = []
            for k in sorted(node):
                v = node[k]
                if isinstance(v, CfgNode):
                    out.append(" " * ind + f"{k}:")
                    out.extend(fmt(v, ind + 2))
                else:
                    out.append(" " * ind + f"{k}: {v}")
            return out
        return "\n".join(fmt(self, 0))


CN = CfgNode


def model(cfg):            # commu/model/config_helper.py:4-15
    cfg.MODEL = CN()
    cfg.MODEL.num_layers = 6
    cfg.MODEL.num_heads = 10
    cfg.MODEL.units = 500
    cfg.MODEL.inner_size = 1000
    cfg.MODEL.dropout = 0.1
    cfg.MODEL.attention_dropout = 0.1
    cfg.MODEL.clamp_len = -1
    cfg.MODEL.same_length = False
    return cfg


def train(cfg):            # commu/model/config_helper.py:18-34
    cfg.TRAIN = CN()
    cfg.TRAIN.batch_size = 256
    cfg.TRAIN.batch_chunk = 4
    cfg.TRAIN.tgt_length = 128
    cfg.TRAIN.mem_length = 1024
    cfg.TRAIN.seed = 1111
    cfg.TRAIN.lr = 0.004
    cfg.TRAIN.lr_min = 0.0001
    cfg.TRAIN.warmup_step = 100
    cfg.TRAIN.clip = 1.0
    cfg.TRAIN.max_step = 20000
    cfg.TRAIN.log_interval = 100
    cfg.TRAIN.eval_interval = 1000
    cfg.TRAIN.weight_decay = 0.0
    return cfg


def init(cfg):             # commu/model/config_helper.py:37-49
    cfg.INITIALIZER = CN()
    cfg.INITIALIZER.base_init = 0.01
    cfg.INITIALIZER.embed_init = 0.01
    cfg.EVALUATE = CN()
    cfg.EVALUATE.batch_size = 10
    cfg.EVALUATE.tgt_length = 128
    cfg.EVALUATE.mem_length = 2048
    return cfg


def get_default_cfg_training():          # commu/model/config_helper.py:52-58
    cfg = CN()
    cfg = init(cfg)
    cfg = model(cfg)
    cfg = train(cfg)
    cfg.freeze()
    return cfg


def get_default_cfg_inference():         # commu/model/config_helper.py:61-80
    cfg = CN()
    cfg.MODEL = CN()
    cfg.MODEL.memory_length = 4146
    cfg.MODEL.device = "gpu"
    cfg.SAMPLING = CN()
    cfg.SAMPLING.threshold = 32.0
    cfg.SAMPLING.temperature = 0.95
    cfg.GENERATION = CN()
    cfg.GENERATION.generation_length = 4096
    cfg.freeze()
    return cfg


def get_cfg(num_layers=6, num_heads=8, units=512, inner_size=1024, tgt_length=1024, mem_length=0,
            batch_size=64, batch_chunk=1, dropout=0.1, attention_dropout=0.1, same_length=False, clamp_len=-1, **train_kw):
    """Reference config with overridden shape fields (the reference has no override mechanism:
    its defaults are hard-coded, config_helper.py:52-58)."""
    cfg = get_default_cfg_training()
    cfg.defrost()
    cfg.MODEL.num_layers, cfg.MODEL.num_heads = num_layers, num_heads
    cfg.MODEL.units, cfg.MODEL.inner_size = units, inner_size
    cfg.MODEL.dropout, cfg.MODEL.attention_dropout = dropout, attention_dropout
    cfg.MODEL.same_length = same_length
    cfg.MODEL.clamp_len = clamp_len
    cfg.TRAIN.tgt_length, cfg.TRAIN.mem_length = tgt_length, mem_length
    cfg.TRAIN.batch_size, cfg.TRAIN.batch_chunk = batch_size, batch_chunk
    for k, v in train_kw.items():
        cfg.TRAIN[k] = v
    cfg.freeze()
    return cfg
