"""Oracle (test infrastructure, see oracle/__init__.py): restatement of the reference's packed-stream batch
iterator, commu/model/dataset.py:117-183, as plain Python loops over columns and batches.

Pinned by fixture G8 (tests/golden/g8_dataset.npz, produced by running the reference itself); used to check the
product's epoch scheduler (commu_amd/model/dataset.py::schedule_epoch) on corpora and epoch counts the fixture does
not reach (empty sequences, many epochs, columns that run dry).
"""
from __future__ import annotations

import numpy as np


def packed_batches(seqs, batch_size, bptt, do_shuffle, seed, max_batches):
    """seqs: list of 1-D int arrays WITH the start token.  Yields (data[T,B], target[T,B], reset[B], ntok)."""
    n = len(seqs)
    lens = [len(s) for s in seqs]
    perm = np.arange(n)                                              # dataset.py:140-142
    rng = np.random.RandomState(seed) if do_shuffle else None
    if do_shuffle:
        rng.shuffle(perm)
    assert batch_size < n                                            # :138
    tracker = [(i, 0) for i in range(batch_size)]                    # :139 (index into perm, position)
    next_index = batch_size
    out = 0
    while out < max_batches:
        data = np.zeros((bptt, batch_size), dtype=np.int64)          # :146-148
        target = np.zeros((bptt, batch_size), dtype=np.int64)
        reset = np.zeros(batch_size, dtype=bool)
        ntok = 0
        for i in range(batch_size):                                  # :150
            idx, pos = tracker[i]
            while idx < n:                                           # :152
                seq_id = perm[idx]
                if pos + 1 >= lens[seq_id]:                          # :154-160
                    idx, pos = next_index, 0
                    tracker[i] = (idx, pos)
                    next_index += 1
                    reset[i] = True
                    continue
                k = min(lens[seq_id] - 1 - pos, bptt)                # :162
                data[:k, i] = seqs[seq_id][pos:pos + k]              # :163-164
                target[:k, i] = seqs[seq_id][pos + 1:pos + 1 + k]
                ntok += k
                tracker[i] = (idx, pos + k)                          # :166
                break
        if ntok == 0:                                                # :170-177
            if not do_shuffle:
                return
            rng.shuffle(perm)
            tracker = [(i, 0) for i in range(batch_size)]
            next_index = batch_size
            continue
        out += 1
        yield data, target, reset, ntok
