"""Decode loop of the hot path: the reference's chord-forced autoregressive generation
(commu/midi_generator/midi_inferrer.py:16-354) with the model step and the sampling step on the
GPU (commu_sample_topk: temperature + softmax + top-k + wrong-token mask + draw in one kernel).

`TeacherForceTask` is the host-side forcing state machine (pure integer logic, same rules and
the same order of checks as the reference, including quirks Q3/Q4/Q5/Q12 of SURVEY.md);
`InferenceTask` keeps the reference's method names.  The draw uses an injected uniform variate
(inverse CDF) because torch.multinomial's RNG stream cannot be reproduced on the device; the
distribution drawn from is the reference's.
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from .. import ops


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


class TeacherForceTask:
    """midi_inferrer.py:16-169.  `input_data` needs `num_measures` and `chord_token_components`."""

    def __init__(self, input_data):
        self.input_data = input_data
        self.next_tokens_forced: List[int] = []
        self.wrong_tokens: List[int] = []
        self.no_sequence_appended = False
        self.is_incomplete = input_data.num_measures % 4 != 0
        self.incomplete_filled = not self.is_incomplete
        comps = input_data.chord_token_components
        self.chord_token = list(comps["chord_token"])
        self.chord_position = list(comps["chord_position"])
        assert len(self.chord_token) == len(self.chord_position), "Wrong Chord Length"
        self.chord_length = len(self.chord_token)
        self.inter_chord_flags = [p != TOKEN_OFFSET.POSITION for p in self.chord_position]

    # -- predicates (:35-114)
    def check_first_position(self, seq):
        return self.incomplete_filled and seq[-1] == TOKEN_OFFSET.BAR

    def check_remnant_chord(self):
        return bool(len(self.chord_token) * len(self.chord_position))

    def check_length_fit(self):
        return self.chord_length == int(self.input_data.num_measures // 4 * 4)

    def check_position_fit(self, seq):
        return seq[-2] == TOKEN_OFFSET.BAR and seq[-1] == TOKEN_OFFSET.POSITION

    def check_one_chord_per_bar_case(self, seq):
        return (self.check_remnant_chord() and self.incomplete_filled and self.check_length_fit()
                and self.check_position_fit(seq))

    def check_mul_chord_per_bar_case(self, seq):
        common = self.check_remnant_chord() and self.incomplete_filled and not self.check_length_fit()
        if not common:
            return False
        if self.check_position_fit(seq):
            return True
        return seq[-1] == self.chord_position[0] and self.inter_chord_flags[0]

    def check_chord_position_passed(self, token):
        if not self.check_remnant_chord():
            return False
        passed = (self.chord_position[0] < token < TOKEN_OFFSET.POSITION + DEFAULT_POSITION_RESOLUTION
                  or token == TOKEN_OFFSET.BAR)
        return self.inter_chord_flags[0] and passed

    def check_wrong_chord_token_generated(self, token):
        return TOKEN_OFFSET.CHORD_START <= token <= TOKEN_OFFSET.CHORD_END

    def check_wrong_eos_generated(self, token):
        return self.check_remnant_chord() and token == TOKEN_OFFSET.EOS

    def check_wrong_bar_token_generated(self, token):
        return not self.check_remnant_chord() and token == TOKEN_OFFSET.BAR

    # -- actions (:116-144)
    def teach_first_position(self):
        self.next_tokens_forced.append(int(TOKEN_OFFSET.POSITION))

    def teach_chord_token(self):
        self.next_tokens_forced.append(self.chord_token.pop(0))
        self.chord_position.pop(0)
        self.inter_chord_flags.pop(0)
        self.wrong_tokens = []

    def teach_chord_position(self):
        self.next_tokens_forced.append(self.chord_position[0])
        self.wrong_tokens = []

    def teach_wrong_chord_token(self, wrong_token):
        self.no_sequence_appended = True
        self.wrong_tokens.append(wrong_token)

    def teach_remnant_chord(self):
        self.next_tokens_forced.append(self.chord_position[0] if self.inter_chord_flags[0] else TOKEN_OFFSET.BAR)

    def teach_eos(self):
        self.next_tokens_forced.append(TOKEN_OFFSET.EOS)

    def validate_teacher_forced_sequence(self, seq) -> None:              # :146-169
        num_bars = seq.count(TOKEN_OFFSET.BAR)
        num_chord = sum(1 for t in seq if TOKEN_OFFSET.CHORD_START <= t <= TOKEN_OFFSET.CHORD_END)
        if len(self.chord_token) != 0:
            raise Exception(f"remnant chord length: {len(self.chord_token)} \nerror in teacher forcing")
        if num_bars != int(math.ceil(self.input_data.num_measures)):
            raise Exception(f"bar length: {num_bars} \nerror in bar length")
        if num_chord != self.chord_length:
            raise Exception(f"num_chord: {num_chord} vs {self.chord_length} \nerror in chord length")


class InferenceTask:
    """midi_inferrer.py:172-354 with the model/sampling steps on the device."""

    def __init__(self, device: torch.device):
        self.device = device
        self.uniform_source = None       # callable -> float in [0,1); default numpy RandomState
        self.trace = None                # optional list collecting (fed token, mlen in, mlen out)

    def __call__(self, model, input_data, inference_cfg):
        self.model = model
        self.input_data = input_data
        self.inference_cfg = inference_cfg

    # -- model steps
    def init_seq_and_mems(self, encoded_meta: List[int], num_conditional_tokens: int):      # :186-197
        seq = [0]
        ctx = torch.tensor(seq + encoded_meta[:num_conditional_tokens - 1], dtype=torch.long,
                           device=self.device)[:, None]
        _, init_mems = self.model.forward_generate(ctx, mems=None)
        return seq + encoded_meta[:num_conditional_tokens], init_mems

    def calc_logits_and_mems(self, seq: List[int], mems):                                    # :199-207
        tok = torch.tensor([[seq[-1]]], dtype=torch.long, device=self.device)
        mlen_in = 0 if mems is None else mems.shape[1]
        all_logits, mems = self.model.forward_generate(tok, mems)
        if self.trace is not None:
            self.trace.append((int(seq[-1]), mlen_in, mems.shape[1]))
        # the reference returns the view all_logits[-1, 0][1:]; the kernel wants the whole row and
        # skips column 0 itself (Q6), so keep the [1, 729] row (fp32, contiguous copy we own)
        return all_logits[-1, 0:1].contiguous(), mems

    # -- sampling step: calc_probs + apply_sampling + infer_token in one kernel (:209-237)
    def sample(self, logits_row, wrong_tokens: Sequence[int], want_probs=False):
        dev = logits_row.device
        wrong = None
        if wrong_tokens:
            w = torch.zeros(1, TOKEN_OFFSET.VOCAB_SIZE, dtype=torch.uint8)
            w[0, list(wrong_tokens)] = 1
            wrong = w.to(dev)
        temp = float(self.input_data.temperature)
        u = None
        if temp != 0:
            u = torch.tensor([self._next_uniform()], dtype=torch.float32, device=dev)
        probs = torch.empty(1, TOKEN_OFFSET.VOCAB_SIZE, device=dev) if want_probs else None
        tok = ops.sample_topk(logits_row, temp, int(self.input_data.top_k), wrong=wrong, uniforms=u, probs_out=probs)
        t = int(tok.item())                      # the one host sync per generated token
        if t < 0:
            raise RuntimeError("invalid multinomial distribution (sum of probabilities <= 0)")
        return (t, probs) if want_probs else t

    def _next_uniform(self):
        if self.uniform_source is None:
            rng = np.random.RandomState(0)
            self.uniform_source = lambda: float(rng.random_sample())
        return self.uniform_source()

    def generate_sequence(self, seq, mems):                                                  # :239-320
        logits = None
        teacher = TeacherForceTask(self.input_data)
        first_loop = True
        for _ in range(self.inference_cfg.GENERATION.generation_length):
            if seq[-1] == 1:
                break
            if teacher.next_tokens_forced:
                seq.append(teacher.next_tokens_forced.pop(0))
                logits, mems = self.calc_logits_and_mems(seq, mems)
                continue
            if teacher.no_sequence_appended:
                assert logits is not None
                teacher.no_sequence_appended = False
            elif first_loop:
                logits, _ = self.calc_logits_and_mems(seq, mems)          # mems discarded (Q3)
                first_loop = False
            else:
                logits, mems = self.calc_logits_and_mems(seq, mems)
            # (the reference computes probs here; the fused kernel computes them when drawing -- the
            #  in-place temperature division must still happen exactly once per pass, see below)
            if not teacher.incomplete_filled:
                teacher.incomplete_filled = seq.count(TOKEN_OFFSET.BAR) > 1
            forced_now = False
            if teacher.check_first_position(seq):
                teacher.teach_first_position()
                forced_now = True
            elif teacher.check_one_chord_per_bar_case(seq):
                teacher.teach_chord_token()
                forced_now = True
            elif teacher.check_mul_chord_per_bar_case(seq):
                teacher.teach_chord_token()
                forced_now = True
            if forced_now:
                # calc_probs already divided the logits in place in the reference (midi_inferrer.py:216,262)
                # -- irrelevant here because the next pass recomputes logits from a model step
                continue
            try:
                token = self.sample(logits, teacher.wrong_tokens)
            except RuntimeError:
                seq = None
                break
            if teacher.check_chord_position_passed(token):
                teacher.teach_chord_position()
                continue
            if teacher.check_wrong_chord_token_generated(token):
                teacher.teach_wrong_chord_token(token)
                continue
            if teacher.check_wrong_eos_generated(token):
                teacher.teach_remnant_chord()
                continue
            if teacher.check_wrong_bar_token_generated(token):
                teacher.teach_eos()
                continue
            seq.append(token)
        self.last_teacher = teacher
        self.last_raw_seq = None if seq is None else list(seq)
        try:
            teacher.validate_teacher_forced_sequence(seq)
        except Exception:
            seq = None
        return seq

    def validate_generated_sequence(self, seq: List[int]) -> bool:                           # :322-336
        num_note = 0
        for idx, token in enumerate(seq):
            if idx + 2 > len(seq) - 1:
                break
            if TOKEN_OFFSET.NOTE_VELOCITY <= token < TOKEN_OFFSET.CHORD_START:
                if (TOKEN_OFFSET.POSITION <= seq[idx - 1] < TOKEN_OFFSET.BPM
                        and TOKEN_OFFSET.PITCH <= seq[idx + 1] < TOKEN_OFFSET.NOTE_VELOCITY
                        and TOKEN_OFFSET.NOTE_DURATION <= seq[idx + 2] < TOKEN_OFFSET.POSITION):
                    num_note += 1
        return num_note > 0

    def execute(self, encoded_meta, max_attempts: Optional[int] = None) -> List[List[int]]:   # :338-354
        num_conditional_tokens = len(encoded_meta)
        idx, attempts = 0, 0
        sequences = []
        while idx != self.input_data.num_generate:
            attempts += 1
            if max_attempts is not None and attempts > max_attempts:
                break
            with torch.no_grad():
                seq, mems = self.init_seq_and_mems(list(encoded_meta), num_conditional_tokens)
                seq = self.generate_sequence(seq, mems)
                if seq is None:
                    continue
                if not self.validate_generated_sequence(seq):
                    continue
            sequences.append(seq)
            idx += 1
        return sequences
