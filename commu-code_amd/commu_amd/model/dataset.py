"""Batch producer of the hot path: the reference's on-disk format and iterators
(commu/model/dataset.py:6-237) plus a synthetic generator for benchmarking.

`ComMUDataset` reads `input_{train,val}.npy` / `target_{train,val}.npy` object arrays (meta: 11
ints per sample, events: int16 ending in EOS=1), prepends the start token 0, and serves
 * get_iterator: packed streams -- B columns of T tokens, per-column `reset_mem` when a column
   moves on to a new sequence, pad 0, count of non-pad target tokens (dataset.py:117-183);
 * eval_iterator: contiguous rank shards, one batch of sequences at a time (dataset.py:185-237).
Host batches are assembled in pinned memory and copied asynchronously.
"""
from __future__ import annotations

import numpy as np
import torch

VOCAB_SIZE = 729      # commu/preprocessor/encoder/event_tokens.py:329


class BaseVocab:
    """commu/model/dataset.py:6-15."""

    def __init__(self):
        self.vec_len = 0

    @property
    def pad_id(self):
        return 0

    def __len__(self):
        return VOCAB_SIZE


def _to_device(t, device):
    dev = torch.device(device)
    if dev.type == "cuda":
        return t.pin_memory().to(dev, non_blocking=True)
    return t.clone()


class ComMUDataset:
    def __init__(self, data_dir, cfg, sequences=None):
        """data_dir: folder with the four .npy files; or pass `sequences` =
        {"train": [1-D int arrays], "valid": [...]} directly (synthetic corpora)."""
        self._vocab = BaseVocab()
        self.cfg = cfg
        if sequences is None:
            sequences = {"train": self.load_cache_data(data_dir, "train"),
                         "valid": self.load_cache_data(data_dir, "valid")}
        pad = self._vocab.pad_id        # pad doubles as the start token (dataset.py:31-37)
        self._data = {}
        for split in ("train", "valid"):
            self._data[split] = [torch.from_numpy(np.insert(np.asarray(a, dtype=np.int64), 0, pad))
                                 for a in sequences[split]]
        self._data["test"] = self._data["valid"]        # dataset.py:81-86: "test" is the validation file
        self._len = {k: np.array([e.shape[0] for e in v], dtype=np.int32) for k, v in self._data.items()}

    @staticmethod
    def load_cache_data(dir_name, mode):
        tag = "train" if mode == "train" else "val"
        data_input = np.load(f"{dir_name}/input_{tag}.npy", allow_pickle=True)
        data_target = np.load(f"{dir_name}/target_{tag}.npy", allow_pickle=True)
        return [np.concatenate((np.array(data_input[i], dtype=int), data_target[i])) for i in range(len(data_input))]

    vocab = property(lambda self: self._vocab)
    train_data = property(lambda self: self._data["train"])
    valid_data = property(lambda self: self._data["valid"])
    test_data = property(lambda self: self._data["test"])
    train_seq_length = property(lambda self: self._len["train"])
    valid_seq_length = property(lambda self: self._len["valid"])
    test_seq_length = property(lambda self: self._len["test"])

    def get_iterator(self, batch_size, bptt, device, split="train", do_shuffle=True, seed=None):
        if split not in self._data:
            raise NotImplementedError
        seqs, lens = self._data[split], self._len[split]
        total = len(seqs)
        pad = self._vocab.pad_id

        def iterator():
            perm = np.arange(total)
            rng = None
            if do_shuffle:
                rng = np.random.RandomState(seed)
                rng.shuffle(perm)
            assert batch_size < total
            cursor = [(i, 0) for i in range(batch_size)]      # (index into perm, position) per column
            next_idx = batch_size
            while True:
                data = torch.full((bptt, batch_size), pad, dtype=torch.long)
                target = torch.full((bptt, batch_size), pad, dtype=torch.long)
                reset = torch.zeros(batch_size, dtype=torch.bool)
                ntok = 0
                for col in range(batch_size):
                    idx, pos = cursor[col]
                    while idx < total:
                        sid = perm[idx]
                        n = lens[sid]
                        if pos + 1 >= n:                      # sequence exhausted: take the next unused one
                            idx, pos = next_idx, 0
                            cursor[col] = (idx, pos)
                            next_idx += 1
                            reset[col] = True
                            continue
                        k = min(n - 1 - pos, bptt)
                        data[:k, col] = seqs[sid][pos:pos + k]
                        target[:k, col] = seqs[sid][pos + 1:pos + 1 + k]
                        ntok += k
                        cursor[col] = (idx, pos + k)
                        break
                if ntok == 0:                                 # epoch end
                    if not do_shuffle:
                        return
                    rng.shuffle(perm)
                    cursor = [(i, 0) for i in range(batch_size)]
                    next_idx = batch_size
                    continue
                yield _to_device(data, device), _to_device(target, device), _to_device(reset, device), ntok

        return iterator

    def eval_iterator(self, batch_size, bptt, device, split="valid", local_rank=0, world_size=0):
        if split not in ("valid", "test"):
            raise NotImplementedError
        seqs, lens = self._data[split], self._len[split]
        if world_size > 0:                                     # contiguous rank shards, dataset.py:196-205
            n_all = len(seqs)
            beg = n_all // world_size * local_rank
            end = n_all if local_rank == world_size - 1 else n_all // world_size * (local_rank + 1)
            seqs, lens = seqs[beg:end], lens[beg:end]
        total = len(seqs)
        pad = self._vocab.pad_id

        def iterator():
            for b0 in range(0, total, batch_size):
                b1 = min(b0 + batch_size, total)
                reset_all = True
                longest = max(lens[b0:b1])
                for s0 in range(0, longest - 1, bptt):
                    data = torch.full((bptt, batch_size), pad, dtype=torch.long)
                    target = torch.full((bptt, batch_size), pad, dtype=torch.long)
                    ntok = 0
                    for i in range(b0, b1):
                        if lens[i] > s0 + 1:
                            k = min(s0 + bptt, lens[i] - 1) - s0
                            data[:k, i - b0] = seqs[i][s0:s0 + k]
                            target[:k, i - b0] = seqs[i][s0 + 1:s0 + k + 1]
                            ntok += k
                    yield _to_device(data, device), _to_device(target, device), reset_all, ntok
                    reset_all = False

        return iterator


def synthetic_batch(bptt, batch_size, device, seed, reset_prob=0.0):
    """Synthetic token batch of the benchmark: ids uniform in [2, 729),
    target = data shifted by one, no pads, optional Bernoulli reset flags."""
    g = torch.Generator().manual_seed(seed)
    stream = torch.randint(2, VOCAB_SIZE, (bptt + 1, batch_size), generator=g)
    reset = torch.rand(batch_size, generator=g) < reset_prob
    return (_to_device(stream[:-1].contiguous(), device), _to_device(stream[1:].contiguous(), device),
            _to_device(reset, device), bptt * batch_size)
