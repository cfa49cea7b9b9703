"""Oracle (test infrastructure, see oracle/__init__.py): restatement of the reference's
sampling step and chord-forced decode loop,
commu/midi_generator/midi_inferrer.py:16-169 (TeacherForceTask) and :186-320
(InferenceTask.init_seq_and_mems / calc_probs / apply_sampling / generate_sequence).

The model step is passed in as a callable so the same loop can be driven by the
oracle model (oracle/xl_ref.py) in the golden tests.  Pinned by fixtures G6/G9
(tests/golden/make_golden.py).
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

# token ranges, commu/preprocessor/encoder/event_tokens.py:308-329
EOS, BAR, CHORD_START, CHORD_END, POSITION = 1, 2, 195, 303, 432
POSITION_RESOLUTION = 128  # commu/preprocessor/utils/constants.py:25


def calc_probs(logits728: torch.Tensor, temperature: float) -> torch.Tensor:
    """midi_inferrer.py:209-221.  ``logits728`` is logits[1:] (pad column dropped, Q6) and is
    divided IN PLACE by the temperature (Q5); returns the 729-wide probability vector."""
    if temperature == 0:
        probs = torch.zeros_like(logits728)
        probs[logits728.argmax()] = 1.0
    else:
        logits728 /= temperature
        probs = F.softmax(logits728, dim=-1)
    return F.pad(probs, [1, 0])


def apply_sampling(probs: torch.Tensor, top_k: int, wrong_tokens: Sequence[int]) -> torch.Tensor:
    """midi_inferrer.py:223-232: keep the top-k entries, zero the rejected chord tokens, renormalise."""
    _, idx = torch.topk(probs, top_k)
    keep = torch.zeros_like(probs)
    keep[idx] = 1.0
    for w in wrong_tokens:
        keep[w] = 0.0
    out = probs * keep
    return out / out.sum()


def apply_top_p(probs: torch.Tensor, top_p: float) -> torch.Tensor:
    """Nucleus filter of the build's extra sampling mode (the reference has top-k only): in order of decreasing
    probability (ties: lowest id first) a token is kept while the mass before it is < top_p; renormalised."""
    if top_p >= 1.0:
        return probs
    srt, idx = torch.sort(probs.double(), descending=True, stable=True)
    before = torch.cumsum(srt, 0) - srt
    keep = torch.zeros_like(probs)
    keep[idx[(before < top_p) & (srt > 0)]] = 1.0
    out = probs * keep
    return out / out.sum()


def draw_inverse_cdf(probs: torch.Tensor, u: float) -> int:
    """Draw with an injected uniform variate: smallest t with cumsum(p)[t] > u.
    (torch.multinomial's RNG stream, midi_inferrer.py:234-237, is not reproducible on a
    device; the distribution is the same.)"""
    cdf = torch.cumsum(probs.double(), 0)
    t = int(torch.searchsorted(cdf, torch.tensor(u, dtype=torch.float64), right=True))
    return min(t, probs.numel() - 1)


class ForcingState:
    """Chord/bar forcing state machine, midi_inferrer.py:16-169, as plain lists."""

    def __init__(self, chord_token: List[int], chord_position: List[int], num_measures: float):
        assert len(chord_token) == len(chord_position)
        self.tok = list(chord_token)
        self.pos = list(chord_position)
        self.inter = [p != POSITION for p in self.pos]          # :28-33
        self.n_chords = len(self.tok)
        self.num_measures = num_measures
        self.forced: List[int] = []
        self.wrong: List[int] = []
        self.redo = False                                        # no_sequence_appended
        self.filled = (num_measures % 4 == 0)                    # incomplete_filled :22-23

    def remnant(self) -> bool:                                   # :41-46
        return bool(len(self.tok) * len(self.pos))

    def length_fit(self) -> bool:                                # :48-52
        return self.n_chords == int(self.num_measures // 4 * 4)

    def position_fit(self, seq) -> bool:                         # :54-58
        return seq[-2] == BAR and seq[-1] == POSITION

    def chord_due(self, seq) -> bool:                            # :60-90
        if not (self.remnant() and self.filled):
            return False
        if self.length_fit():
            return self.position_fit(seq)
        if self.position_fit(seq):
            return True
        return seq[-1] == self.pos[0] and self.inter[0]

    def take_chord(self):                                        # :122-127
        self.forced.append(self.tok.pop(0))
        self.pos.pop(0)
        self.inter.pop(0)
        self.wrong = []

    def position_passed(self, token: int) -> bool:               # :92-102
        if not self.remnant():
            return False
        passed = (self.pos[0] < token < POSITION + POSITION_RESOLUTION) or token == BAR
        return self.inter[0] and passed


def generate_sequence(step: Callable[[int, object], Tuple[torch.Tensor, object]],
                      seq: List[int], mems, *, chord_token: List[int], chord_position: List[int],
                      num_measures: float, temperature: float, top_k: int,
                      uniforms: Optional[Sequence[float]] = None, max_iters: int = 4096,
                      trace: Optional[list] = None) -> Optional[List[int]]:
    """midi_inferrer.py:239-313 (loop body) -- ``step(token, mems) -> (logits[729], new_mems)``.

    ``uniforms`` supplies the variates for the inverse-CDF draws (greedy needs none).
    ``trace`` collects (fed_token, kept) per model call, pinning quirks Q3/Q4."""
    fs = ForcingState(chord_token, chord_position, num_measures)
    logits = None
    first = True
    n_draw = 0

    def call(tok, m, keep=True):
        lg, nm = step(tok, m)
        if trace is not None:
            trace.append((int(tok), bool(keep)))
        return lg[1:].clone(), (nm if keep else m)

    for _ in range(max_iters):
        if seq[-1] == EOS:
            break
        if fs.forced:                                            # :247-251
            seq.append(fs.forced.pop(0))
            logits, mems = call(seq[-1], mems)
            continue
        if fs.redo:                                              # :253-255 (Q5: reuse divided logits)
            fs.redo = False
        elif first:                                              # :256-258 (Q3: mems discarded)
            logits, mems = call(seq[-1], mems, keep=False)
            first = False
        else:                                                    # :259-260 (Q4)
            logits, mems = call(seq[-1], mems)
        probs = apply_sampling(calc_probs(logits, temperature), top_k, fs.wrong)
        if not fs.filled:                                        # :267-268
            fs.filled = seq.count(BAR) > 1
        if fs.filled and seq[-1] == BAR:                         # :271-273
            fs.forced.append(POSITION)
            continue
        if fs.chord_due(seq):                                    # :276-283
            fs.take_chord()
            continue
        if temperature == 0:
            if not torch.isfinite(probs).all():                  # Q12: multinomial would raise
                return None
            token = int(probs.argmax())
        else:
            u = uniforms[n_draw]
            n_draw += 1
            token = draw_inverse_cdf(probs, u)
        if fs.position_passed(token):                            # :294-296
            fs.forced.append(fs.pos[0])
            fs.wrong = []
            continue
        if CHORD_START <= token <= CHORD_END:                    # :299-301
            fs.redo = True
            fs.wrong.append(token)
            continue
        if fs.remnant() and token == EOS:                        # :304-306
            fs.forced.append(fs.pos[0] if fs.inter[0] else BAR)
            continue
        if (not fs.remnant()) and token == BAR:                  # :309-311
            fs.forced.append(EOS)
            continue
        seq.append(token)
    return seq
