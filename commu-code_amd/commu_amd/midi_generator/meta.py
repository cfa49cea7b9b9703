"""Host-side front end of the generation path: meta tokens and chord-forcing components.

Restates, as data tables + two small functions, what the reference spreads over
commu/preprocessor/encoder/meta.py:12-250, commu/preprocessor/utils/constants.py:22-176,
commu/preprocessor/encoder/encoder_utils.py:59-182,356-368 and
commu/midi_generator/container.py:17-61 / info_preprocessor.py:9-44, so that the decode loop
(`midi_inferrer.InferenceTask`, `generate.BatchedGenerator`) can be driven from the reference's own argument
dictionary without miditoolkit / pydantic.  Integer logic only; pinned by tests/golden/g7_meta*.
MIDI writing stays out of scope.
"""
from __future__ import annotations

import math
from fractions import Fraction
from typing import Dict, List, Sequence

UNKNOWN = "unknown"

# ---- field order of the 11 meta tokens and, per field, (token of "unknown", offset added to a known value)
META_FIELDS = ("bpm", "audio_key", "time_signature", "pitch_range", "num_measures", "inst", "genre", "min_velocity",
               "max_velocity", "track_role", "rhythm")
_BASE = {"bpm": 560, "audio_key": 601, "time_signature": 626, "pitch_range": 630, "num_measures": 638, "inst": 641,
         "genre": 650, "velocity": 653, "track_role": 719, "rhythm": 726}          # event_tokens.py:316-326

_NOTE_NAMES = ["c", "c#", "d", "d#", "e", "f", "f#", "g", "g#", "a", "a#", "b"]
_FLAT_OF = {"c#": "db", "d#": "eb", "f#": "gb", "g#": "ab", "a#": "bb"}
KEY_MAP = {}
for _mode, _shift in (("major", 0), ("minor", 12)):
    for _i, _n in enumerate(_NOTE_NAMES):
        KEY_MAP[_n + _mode] = _i + _shift
        if _n in _FLAT_OF:
            KEY_MAP[_FLAT_OF[_n] + _mode] = _i + _shift
TIME_SIG_MAP = {"4/4": 0, "3/4": 1, "6/8": 2, "12/8": 3}
PITCH_RANGE_MAP = {n: i for i, n in enumerate(["very_low", "low", "mid_low", "mid", "mid_high", "high", "very_high"])}
GENRE_MAP = {"newage": 0, "cinematic": 1}
TRACK_ROLE_MAP = {n: i for i, n in enumerate(["main_melody", "sub_melody", "accompaniment", "bass", "pad", "riff"])}
RHYTHM_MAP = {"standard": 0, "triplet": 1}
# instrument families (constants.py:97-160): family id -> member names
_INST_FAMILIES = {
    0: "acoustic_piano electric_piano harpsichord keyboard organ",
    1: "accordion synth_lead",
    2: "bell celesta glockenspiel marimba synth_bell vibraphone xylophone orgel",
    3: "acoustic_bass acoustic_guitar banjo electric_bass electric_guitar_clean electric_guitar_distortion harp mandolin "
       "nylon_guitar oud sitar synth_bass synth_bass_808 synth_bass_wobble ukulele zither yanggeum",
    4: "fiddle pad_synth string_cello string_double_bass string_ensemble string_viola string_violin synth_pad",
    5: "bassoon brass_ensemble clarinet flute horn oboe recorder trombone trumpet tuba synth_brass sax bamboo_flute",
    6: "drums_full drums_tops percussion timpani",
    7: "choir synth_pluck synth_voice whistle",
    8: "vocal",
}
INST_MAP = {name: fam for fam, names in _INST_FAMILIES.items() for name in names.split()}
_MAPS = {"audio_key": KEY_MAP, "time_signature": TIME_SIG_MAP, "pitch_range": PITCH_RANGE_MAP, "inst": INST_MAP,
         "genre": GENRE_MAP, "track_role": TRACK_ROLE_MAP, "rhythm": RHYTHM_MAP}


class UnprocessableMidiError(Exception):
    """commu/preprocessor/utils/exceptions.py (name kept so callers can catch the same error)."""


def _get(meta, name):
    return meta[name] if isinstance(meta, dict) else getattr(meta, name)


def encode_meta(meta) -> List[int]:
    """The 11 meta tokens (meta.py:226-240).  `meta`: dict or object with the MidiMeta fields."""
    out = []
    for field in META_FIELDS:
        v = _get(meta, field)
        base = _BASE["velocity" if field.endswith("velocity") else field]
        if field == "num_measures":
            if v == UNKNOWN:
                raise UnprocessableMidiError("num_measures unknown")
            n = math.floor(v)
            slot = {4: 0, 5: 0, 8: 1, 9: 1, 16: 2, 17: 2}.get(n)
            if slot is None:
                raise UnprocessableMidiError(f"num measures ValueError: {n}")
            out.append(base + slot)
            continue
        if v == UNKNOWN:
            out.append(base)
            continue
        if field == "bpm":
            out.append(base + max(1, min(v, 200) // 5))            # offset of bpm is the base itself (meta.py:52)
            continue
        if field == "min_velocity":
            code = math.floor(v / 2)
        elif field == "max_velocity":
            code = math.ceil(v / 2)
        else:
            try:
                code = _MAPS[field][v]
            except KeyError:
                raise UnprocessableMidiError(f"{field} KeyError: {v}")
        out.append(code if code == base else code + base + 1)      # (a value equal to the unknown token is kept)
    return out


# ---- chord vocabulary: token = 195 + 9 * root + quality, "NN" = 303 (event_tokens.py:313-314 and the event list)
_ROOTS = ["a", "a#", "b", "c", "c#", "d", "d#", "e", "f", "f#", "g", "g#"]
_QUALS = ["", "7", "+", "dim", "m", "m7", "m7b5", "maj7", "sus4"]
# spelling -> vocabulary quality.  Flats accept the long list (encoder_utils.py:61-148), natural roots the
# shorter one (:150-182, where madd2 / mM7 fold to m7, not m); sharps only the nine base qualities.
_FLAT_ALIAS = {"": "", "maj": "", "6": "", "maj7": "maj7", "add2": "maj7", "sus2": "maj7", "7": "7", "dim": "dim",
               "dim7": "dim", "+": "+", "m": "m", "m6": "m", "mM7": "m", "m7": "m7", "madd2": "m7", "sus4": "sus4",
               "7sus4": "sus4", "m7b5": "m7b5"}
_NATURAL_ALIAS = {"7sus4": "sus4", "m6": "m", "sus2": "maj7", "add2": "maj7", "6": "", "dim7": "dim", "madd2": "m7",
                  "mM7": "m7"}
_FLAT_ROOT = {"ab": "g#", "bb": "a#", "db": "c#", "eb": "d#", "gb": "f#"}


def _chord_vocab() -> Dict[str, int]:
    v = {r + q: 195 + 9 * i + j for i, r in enumerate(_ROOTS) for j, q in enumerate(_QUALS)}
    v["NN"] = 303
    for flat, sharp in _FLAT_ROOT.items():
        for spelled, q in _FLAT_ALIAS.items():
            v[flat + spelled] = v[sharp + q]
    for r in "abcdefg":
        for spelled, q in _NATURAL_ALIAS.items():
            v[r + spelled] = v[r + q]
    return v


CHORD_VOCAB = _chord_vocab()
POSITION_BASE, POSITION_RESOLUTION = 432, 128


def chord_token_components(chord_progression: Sequence[str], time_signature: str) -> Dict[str, List[int]]:
    """container.py:37-61: one (token, position) per chord CHANGE (and per bar start).  The position of a change
    inside a bar reproduces the reference's decimal-string arithmetic (e.g. 1/6 of a bar -> 432 + 21)."""
    per_bar = int(Fraction(time_signature) * 4) * 2                  # chords per bar: one per half beat
    n_bars = int(len(chord_progression) / per_bar)
    # numpy.array_split semantics: the first len % n_bars bars get one more element
    q, r = divmod(len(chord_progression), n_bars)
    tokens, positions, last, start = [], [], None, 0
    for bar in range(n_bars):
        size = q + (1 if bar < r else 0)
        for c_idx, chord in enumerate(chord_progression[start:start + size]):
            chord = chord.lower()
            if c_idx == 0 or chord != last:
                where = bar + c_idx / per_bar
                frac = str(where).split(".")[-1]
                positions.append(int(POSITION_BASE + float(frac) * POSITION_RESOLUTION / 10 ** len(frac)))
                tokens.append(CHORD_VOCAB[chord.split("/")[0].split("(")[0]])       # KeyError like the reference
                last = chord
        start += size
    return {"chord_token": tokens, "chord_position": positions}


class InputData:
    """TransXlInputData (container.py:17-64) without pydantic: the fields the decode loop reads."""

    def __init__(self, **kw):
        prog = kw["chord_progression"]
        self.chord_progression = prog.split("-") if isinstance(prog, str) else list(prog)
        for f in META_FIELDS:
            setattr(self, f, kw[f])
        self.num_generate = int(kw.get("num_generate", 1))
        self.top_k = int(kw.get("top_k", 32))
        top_p = kw.get("top_p")                              # not in the reference: nucleus filter after top-k; 1.0 = off
        self.top_p = 1.0 if top_p is None else float(top_p)
        if not 0.0 < self.top_p <= 1.0:
            raise ValueError("top_p must be in (0, 1]")
        self.temperature = float(kw.get("temperature", 0.95))
        self.output_dir = kw.get("output_dir")
        nm = self.num_measures
        if (nm - (nm % 4)) * Fraction(self.time_signature) * 8 != len(self.chord_progression):
            raise ValueError("num_measures not matched with chord progression length")

    @property
    def chord_token_components(self) -> Dict[str, List[int]]:
        return chord_token_components(self.chord_progression, self.time_signature)


class PreprocessTask:
    """info_preprocessor.py:21-44."""

    def __init__(self):
        self.input_data = None

    def get_meta_info_length(self):
        return len(META_FIELDS)

    def normalize_input_data(self, input_data: dict):
        self.input_data = InputData(**input_data)

    def execute(self, input_data: dict) -> List[int]:
        if self.input_data is None:
            self.normalize_input_data(input_data)
        return encode_meta(self.input_data)
