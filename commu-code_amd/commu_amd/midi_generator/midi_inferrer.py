"""Decode loop of the hot path behind the reference's task API (commu/midi_generator/midi_inferrer.py).

The reference generates one sequence at a time with a Python loop around `forward_generate`, a sampling step
and the chord-forcing checks of `TeacherForceTask` (:16-169, :239-320).  Here the loop runs on the device for
all requested sequences at once (commu_amd.generate.ForcedDecoder: K/V-cache decode step, sampling kernel and the
forcing rules as two state-transition kernels, one hipGraph replay per iteration).  This module keeps what the
callers of the reference touch:

  * `TOKEN_OFFSET`            -- the token ranges the rules are written in (event_tokens.py:308-329)
  * `ForcingReport`           -- the outcome of the forcing for one sequence + the reference's structural check
                                 `validate_teacher_forced_sequence` (:146-169)
  * `InferenceTask`           -- `__call__(model, input_data, inference_cfg)` / `execute(encoded_meta)` (:172-184,
                                 :338-354) and `validate_generated_sequence` (:322-336)
"""
from __future__ import annotations

import math
from typing import List, Optional

import torch


class TOKEN_OFFSET:
    """commu/preprocessor/encoder/event_tokens.py:308-329 (values only)."""
    EOS = 1
    BAR = 2
    PITCH = 3
    NOTE_VELOCITY = 131
    CHORD_START = 195
    CHORD_END = 303
    NOTE_DURATION = 304
    POSITION = 432
    BPM = 560
    VOCAB_SIZE = 729


DEFAULT_POSITION_RESOLUTION = 128      # commu/preprocessor/utils/constants.py:25

# token classes of one note event, in sequence order: POSITION, VELOCITY, PITCH, DURATION (sequence_postprocessor.py:34-46)
_NOTE_PATTERN = ((TOKEN_OFFSET.POSITION, TOKEN_OFFSET.BPM), (TOKEN_OFFSET.NOTE_VELOCITY, TOKEN_OFFSET.CHORD_START),
                 (TOKEN_OFFSET.PITCH, TOKEN_OFFSET.NOTE_VELOCITY), (TOKEN_OFFSET.NOTE_DURATION, TOKEN_OFFSET.POSITION))


def count_notes(seq: List[int]) -> int:
    """Number of well-formed note events: a velocity token preceded by a position and followed by pitch, duration
    (the scan of validate_generated_sequence, :322-336, which stops two tokens before the end)."""
    n = 0
    for i in range(1, len(seq) - 2):
        window = (seq[i - 1], seq[i], seq[i + 1], seq[i + 2])
        if all(lo <= t < hi for t, (lo, hi) in zip(window, _NOTE_PATTERN)):
            n += 1
    return n


class ForcingReport:
    """What the forcing state machine was asked to place in one sequence and how far it got."""

    def __init__(self, n_chords: int, num_measures: float):
        self.n_chords, self.num_measures = n_chords, num_measures
        self.length_fit = n_chords == int(num_measures // 4 * 4)          # :48-52 (one chord per bar)
        self.consumed = 0                                                   # chords handed to the sequence

    def validate_teacher_forced_sequence(self, seq: List[int]) -> None:    # :146-169
        bars = seq.count(TOKEN_OFFSET.BAR)
        chords = sum(1 for t in seq if TOKEN_OFFSET.CHORD_START <= t <= TOKEN_OFFSET.CHORD_END)
        if self.consumed != self.n_chords:
            raise Exception(f"remnant chord length: {self.n_chords - self.consumed} \nerror in teacher forcing")
        if bars != int(math.ceil(self.num_measures)):
            raise Exception(f"bar length: {bars} \nerror in bar length")
        if chords != self.n_chords:
            raise Exception(f"num_chord: {chords} vs {self.n_chords} \nerror in chord length")


class InferenceTask:
    """midi_inferrer.py:172-354.  `execute` returns `input_data.num_generate` sequences that passed both of the
    reference's validators; rejected attempts are retried like the reference does (in parallel rounds)."""

    def __init__(self, device: torch.device):
        self.device = device
        self.uniform_seed = 0
        self.attempts = 0

    def __call__(self, model, input_data, inference_cfg):
        self.model = model
        self.input_data = input_data
        self.inference_cfg = inference_cfg

    def validate_generated_sequence(self, seq: List[int]) -> bool:                           # :322-336
        return count_notes(seq) > 0

    def execute(self, encoded_meta, max_rounds: Optional[int] = None) -> List[List[int]]:    # :338-354
        from ..generate import BatchedGenerator
        data = self.input_data
        glen = self.inference_cfg.GENERATION.generation_length
        mlen = getattr(self.inference_cfg.MODEL, "memory_length", 4146) if hasattr(self.inference_cfg, "MODEL") else 4146
        gen = BatchedGenerator(self.model, self.device, glen, mlen)

        def accept(seq, rep) -> bool:
            self.attempts += 1
            if seq is None:
                return False
            try:
                rep.validate_teacher_forced_sequence(seq)
            except Exception:
                return False
            return self.validate_generated_sequence(seq)
        # attempts decoded in up to 64 slots, a finished slot re-armed with the next attempt (the reference tries one
        # sequence after the other until num_generate passed; `max_rounds` bounds the attempts at max_rounds x num_generate)
        out, _ = gen.generate_stream(list(encoded_meta), data, data.temperature, data.top_k, data.num_generate, accept,
                                     top_p=getattr(data, "top_p", 1.0), seed=self.uniform_seed,
                                     max_attempts=None if max_rounds is None else max_rounds * data.num_generate)
        return out
