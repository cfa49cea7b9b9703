"""Batch producer of the hot path: the reference's on-disk format and batch semantics
(commu/model/dataset.py:6-237) behind a packer built for a fast consumer.

What the reference's iterators produce is kept exactly (pinned by the g8_dataset fixture):
 * get_iterator (dataset.py:117-183): B columns of T tokens; a column streams one sequence T tokens at a time
   (target = input shifted by one), takes the next unused sequence of the (shuffled) corpus when its own is
   exhausted -- `reset_mem[col]` is raised in that batch and the column restarts at position 0 --, pads with 0, and
   every batch carries its count of non-pad targets.  An epoch ends with the first all-pad batch.
 * eval_iterator (dataset.py:185-237): contiguous rank shards, B sequences at a time, T-token segments until the
   longest sequence of the group is consumed.

How it is produced is different:
 * the corpus is ONE flat token array + offsets (CSR), not a list of tensors;
 * a whole epoch is SCHEDULED up front: which (sequence, position, length) lands in which (batch, column) follows
   from a list-scheduling pass over the sequences (a column is a machine, a sequence a job of ceil((n-1)/T) batches,
   the next sequence goes to the column that frees up first, lowest column on ties) -- one heap operation per
   sequence instead of the reference's Python loop over every column of every batch;
 * a batch is assembled by one vectorised gather into PINNED host buffers by a worker thread, two batches ahead of
   the consumer, and reaches the GPU by an asynchronous copy.
"""
from __future__ import annotations

import heapq
import queue
import threading

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


class _Corpus:
    """Flat storage of one split: tokens[offsets[s] : offsets[s + 1]] is sequence s (start token 0 included)."""

    def __init__(self, arrays, pad):
        lens = np.array([len(a) + 1 for a in arrays], dtype=np.int64)
        self.offsets = np.ascontiguousarray(np.concatenate([[0], np.cumsum(lens)]), dtype=np.int64)
        self.tokens = np.full(int(self.offsets[-1]) + 1, pad, dtype=np.int64)          # (+1: gathers may touch one past the end)
        for s, a in enumerate(arrays):
            self.tokens[self.offsets[s] + 1:self.offsets[s + 1]] = np.asarray(a, dtype=np.int64)
        self.lens = lens.astype(np.int32)

    def __len__(self):
        return len(self.lens)

    def sequence(self, s):
        return torch.from_numpy(self.tokens[self.offsets[s]:self.offsets[s + 1]])


def schedule_epoch(lens, order, batch_size, bptt):
    """Column schedule of one epoch.  Returns int64 arrays [n_batches, batch_size]: `seq` (sequence id, -1: nothing),
    `pos` (first input position) and `cnt` (tokens), and bool `reset`.

    Sequence order[c] starts in column c; every later sequence goes to the column whose current sequence runs out
    first (lowest column first among those that run out in the same batch), exactly the order in which the
    reference's per-batch / per-column loop hands out `next_idx` (dataset.py:152-166).  Sequences of one token have
    nothing to predict and are skipped on the spot (`pos + 1 >= n`), as there."""
    total = len(order)
    assert batch_size < total                                     # dataset.py:138
    dur = (np.maximum(lens[order].astype(np.int64) - 1, 0) + bptt - 1) // bptt          # batches per sequence
    jobs = [[] for _ in range(batch_size)]                        # per column: (first batch, index into order, restarted?)
    free = []                                                     # (batch at which the column needs a new sequence, column)
    for c in range(batch_size):
        jobs[c].append((0, c, False))
        heapq.heappush(free, (int(dur[c]), c))
    nxt = batch_size
    tail = []                                                     # columns that found the corpus exhausted: (batch, column)
    while free:
        t, c = heapq.heappop(free)
        while nxt < total and dur[nxt] == 0:                      # empty sequences are passed over
            nxt += 1
        if nxt >= total:
            tail.append((t, c))
            continue
        jobs[c].append((t, nxt, True))
        heapq.heappush(free, (t + int(dur[nxt]), c))
        nxt += 1
    n_batches = max(t for t, _ in tail)
    seq = np.full((n_batches, batch_size), -1, dtype=np.int64)
    pos = np.zeros((n_batches, batch_size), dtype=np.int64)
    cnt = np.zeros((n_batches, batch_size), dtype=np.int64)
    reset = np.zeros((n_batches, batch_size), dtype=bool)
    for c in range(batch_size):
        for t0, j, restarted in jobs[c]:
            d = int(dur[j])
            if d == 0:
                continue
            s, n = int(order[j]), int(lens[order[j]])
            k = np.arange(d, dtype=np.int64)
            seq[t0:t0 + d, c] = s
            pos[t0:t0 + d, c] = k * bptt
            cnt[t0:t0 + d, c] = np.minimum(n - 1 - k * bptt, bptt)
            if restarted:
                reset[t0, c] = True
    for t, c in tail:                                             # the batch in which a column finds nothing left still
        if t < n_batches:                                         # raises its flag (dataset.py:158-164)
            reset[t, c] = True
    return seq, pos, cnt, reset


def gather_batch(corpus, seq, pos, cnt, bptt, data, target, pad):
    """data[:cnt[c], c] = sequence seq[c] from pos[c]; target = the same shifted by one; pad elsewhere.
    ONE native call (commu_pack_batch, csrc/pack.hip -- host code of the C-ABI library), which runs without the GIL:
    `data` / `target` are preallocated C-contiguous [bptt, B] int64 numpy views."""
    from .. import _lib
    seq = np.ascontiguousarray(seq, dtype=np.int64)
    pos = np.ascontiguousarray(pos, dtype=np.int64)
    cnt = np.ascontiguousarray(cnt, dtype=np.int64)
    assert data.flags["C_CONTIGUOUS"] and target.flags["C_CONTIGUOUS"] and data.dtype == np.int64 == target.dtype
    assert data.shape == (bptt, len(seq)) == target.shape and corpus.tokens.dtype == np.int64
    n = _lib.call("commu_pack_batch", corpus.tokens.ctypes.data, corpus.offsets.ctypes.data, seq.ctypes.data,
                  pos.ctypes.data, cnt.ctypes.data, len(seq), bptt, int(pad), data.ctypes.data, target.ctypes.data)
    if n < 0:
        raise _lib.CommuHipError(f"commu_pack_batch failed with status {n}")
    return int(n)


class _Prefetcher:
    """A worker thread fills a small ring of (pinned) host batches AND ships them: every batch goes to the GPU by an
    asynchronous copy on a dedicated COPY STREAM, issued by the worker itself, two batches ahead of the consumer.
    (Copies enqueued on the training stream -- the first version -- only run once that stream has drained the step in
    front of them: the ring then recycles at the pace of the GPU and the training loop waits for its own queue.)  The
    consumer makes its stream wait for the copy's event; a pinned slot is refilled once its copy has completed."""

    DEPTH = 3

    def __init__(self, produce, shapes, device):
        self.device = torch.device(device)
        self.cuda = self.device.type == "cuda"
        self.slots = []
        for _ in range(self.DEPTH):
            bufs = [torch.empty(shape, dtype=dtype, pin_memory=self.cuda) for shape, dtype in shapes]
            self.slots.append({"bufs": bufs, "event": None})
        self.ready = queue.Queue(maxsize=self.DEPTH - 1)
        self.stop = False
        self.copy_stream = torch.cuda.Stream(device=self.device) if self.cuda else None
        self.thread = threading.Thread(target=self._work, args=(produce,), daemon=True)
        self.thread.start()

    def _work(self, produce):
        try:
            if self.cuda:
                torch.cuda.set_device(self.device)
            i = 0
            while not self.stop:
                slot = self.slots[i]
                i = (i + 1) % self.DEPTH
                if slot["event"] is not None:
                    slot["event"].synchronize()                   # the copy out of this slot has finished
                extra = produce(slot["bufs"])
                if extra is None:                                 # end of the stream
                    self._put(None)
                    return
                if self.cuda:
                    with torch.cuda.stream(self.copy_stream):
                        out = [b.to(self.device, non_blocking=True) for b in slot["bufs"]]
                        ev = torch.cuda.Event()
                        ev.record(self.copy_stream)
                    slot["event"] = ev
                else:
                    out, ev = [b.clone() for b in slot["bufs"]], None
                if not self._put((out, ev, extra)):
                    return
        except BaseException as exc:                              # surfaced in the consumer
            self._put(exc)

    def _put(self, item):
        """ready.put that gives up when the iterator was closed (a worker blocked on a full queue would otherwise
        never wake: an abandoned iterator leaked its thread and its pinned buffers)."""
        while not self.stop:
            try:
                self.ready.put(item, timeout=0.1)
                return True
            except queue.Full:
                continue
        return False

    def __iter__(self):
        return self

    def __next__(self):
        item = self.ready.get()
        if item is None:
            raise StopIteration
        if isinstance(item, BaseException):
            raise item
        out, ev, extra = item
        if ev is not None:
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(ev)
            for t in out:
                t.record_stream(cur)                              # allocated on the copy stream, used on this one
        return out, extra

    def close(self):
        self.stop = True
        try:                                                      # unblock a worker waiting on a full queue
            while True:
                self.ready.get_nowait()
        except Exception:                                         # queue.Empty (may already be torn down at exit)
            pass
        if self.thread is not threading.current_thread():
            self.thread.join(timeout=2.0)


class ComMUDataset:
    def __init__(self, data_dir, cfg, sequences=None):
        """data_dir: folder with input_/target_{train,val}.npy (meta: 11 ints per sample, events: int16 ending in
        EOS = 1; commu/preprocessor/preprocessor.py:161-162); or pass `sequences` =
        {"train": [1-D int arrays], "valid": [...]} directly (synthetic corpora)."""
        self._vocab = BaseVocab()
        self.cfg = cfg
        if sequences is None:
            sequences = {"train": self.load_cache_data(data_dir, "train"),
                         "valid": self.load_cache_data(data_dir, "valid")}
        pad = self._vocab.pad_id                                  # pad doubles as the start token (dataset.py:31-37)
        self._corpus = {split: _Corpus(sequences[split], pad) for split in ("train", "valid")}
        self._corpus["test"] = self._corpus["valid"]              # dataset.py:81-86: "test" is the validation file

    @staticmethod
    def load_cache_data(dir_name, mode):
        tag = "train" if mode == "train" else "val"
        data_input = np.load(f"{dir_name}/input_{tag}.npy", allow_pickle=True)
        data_target = np.load(f"{dir_name}/target_{tag}.npy", allow_pickle=True)
        return [np.concatenate((np.array(data_input[i], dtype=int), data_target[i])) for i in range(len(data_input))]

    vocab = property(lambda self: self._vocab)
    train_seq_length = property(lambda self: self._corpus["train"].lens)
    valid_seq_length = property(lambda self: self._corpus["valid"].lens)
    test_seq_length = property(lambda self: self._corpus["test"].lens)

    def _as_list(self, split):
        c = self._corpus[split]
        return [c.sequence(s) for s in range(len(c))]

    train_data = property(lambda self: self._as_list("train"))
    valid_data = property(lambda self: self._as_list("valid"))
    test_data = property(lambda self: self._as_list("test"))

    def get_iterator(self, batch_size, bptt, device, split="train", do_shuffle=True, seed=None):
        """Returns a callable that starts the (endless, when shuffling) batch stream:
        (data [T,B] int64, target [T,B] int64, reset_mem [B] bool, number of non-pad targets)."""
        if split not in self._corpus:
            raise NotImplementedError
        corpus = self._corpus[split]
        pad = self._vocab.pad_id

        def iterator():
            rng = np.random.RandomState(seed) if do_shuffle else None
            order = np.arange(len(corpus))
            state = {"t": 0, "plan": None}

            def produce(bufs):
                while state["plan"] is None or state["t"] >= state["plan"][0].shape[0]:
                    if state["plan"] is not None and not do_shuffle:
                        return None                               # one pass over the split
                    if do_shuffle:
                        rng.shuffle(order)                        # in place, epoch after epoch (dataset.py:142,175)
                    state["plan"], state["t"] = schedule_epoch(corpus.lens, order, batch_size, bptt), 0
                seq, pos, cnt, reset = state["plan"]
                t = state["t"]
                ntok = gather_batch(corpus, seq[t], pos[t], cnt[t], bptt, bufs[0].numpy(), bufs[1].numpy(), pad)
                bufs[2].numpy()[:] = reset[t]
                state["t"] = t + 1
                return ntok
            shapes = [((bptt, batch_size), torch.long), ((bptt, batch_size), torch.long), ((batch_size,), torch.bool)]
            pf = _Prefetcher(produce, shapes, device)
            try:
                for (data, target, reset), ntok in pf:
                    yield data, target, reset, ntok
            finally:
                pf.close()

        return iterator

    def eval_iterator(self, batch_size, bptt, device, split="valid", local_rank=0, world_size=0):
        """(data, target, first segment of a new group?, non-pad targets) over this rank's contiguous shard."""
        if split not in ("valid", "test"):
            raise NotImplementedError
        corpus = self._corpus[split]
        lo, hi = 0, len(corpus)
        if world_size > 0:                                        # contiguous rank shards, dataset.py:196-205
            share = len(corpus) // world_size
            lo = share * local_rank
            hi = len(corpus) if local_rank == world_size - 1 else share * (local_rank + 1)
        pad = self._vocab.pad_id

        def iterator():
            # segment list of the whole shard: (first sequence of the group, segment start, first segment?)
            plan = []
            for b0 in range(lo, hi, batch_size):
                longest = int(corpus.lens[b0:min(b0 + batch_size, hi)].max())
                plan += [(b0, s0, s0 == 0) for s0 in range(0, longest - 1, bptt)]
            state = {"i": 0}

            def produce(bufs):
                if state["i"] >= len(plan):
                    return None
                b0, s0, first = plan[state["i"]]
                state["i"] += 1
                ids = np.arange(b0, b0 + batch_size)
                seq = np.where(ids < hi, ids, -1)
                n = np.where(seq >= 0, corpus.lens[np.minimum(ids, len(corpus) - 1)], 0).astype(np.int64)
                cnt = np.clip(n - 1 - s0, 0, bptt)
                seq = np.where(cnt > 0, seq, -1)
                ntok = gather_batch(corpus, seq, np.full(batch_size, s0, dtype=np.int64), cnt, bptt,
                                    bufs[0].numpy(), bufs[1].numpy(), pad)
                return first, ntok
            shapes = [((bptt, batch_size), torch.long), ((bptt, batch_size), torch.long)]
            pf = _Prefetcher(produce, shapes, device)
            try:
                for (data, target), (first, ntok) in pf:
                    yield data, target, first, ntok
            finally:
                pf.close()

        return iterator


def _to_device(t, device):
    dev = torch.device(device)
    if dev.type == "cuda":
        return t.pin_memory().to(dev, non_blocking=True)
    return t.clone()


def synthetic_batch(bptt, batch_size, device, seed, reset_prob=0.0):
    """Synthetic token batch of the benchmark: ids uniform in [2, 729),
    target = data shifted by one, no pads, optional Bernoulli reset flags."""
    g = torch.Generator().manual_seed(seed)
    stream = torch.randint(2, VOCAB_SIZE, (bptt + 1, batch_size), generator=g)
    reset = torch.rand(batch_size, generator=g) < reset_prob
    return (_to_device(stream[:-1].contiguous(), device), _to_device(stream[1:].contiguous(), device),
            _to_device(reset, device), bptt * batch_size)
