"""Token-level generation pipeline: the reference's MidiGenerationPipeline (commu/midi_generator/
generate_pipeline.py, driven by generate.py:53-75) from the argument dictionary to VALIDATED token sequences:

    PreprocessTask (meta tokens, chord components)  ->  `num_generate` sequences decoded IN PARALLEL with the
    K/V-cache decode step and per-sequence chord forcing (generate.BatchedGenerator)  ->  the reference's two
    validators (midi_inferrer.py:146-169, 322-336)  ->  retry the rejected ones (execute, :338-354).

Decoding tokens into a MIDI file (PostprocessTask, miditoolkit) is out of scope: the returned lists are what
sequence_postprocessor.py:34-46 takes.
"""
from __future__ import annotations

from typing import List, Optional

import torch

from ..generate import BatchedGenerator
from .meta import PreprocessTask
from .midi_inferrer import InferenceTask


class TokenGenerationPipeline:
    def __init__(self, model, device: torch.device, generation_length: int = 4096, memory_length: int = 4146):
        self.model, self.device = model, device
        self.generation_length, self.memory_length = generation_length, memory_length
        self.preprocess_task = PreprocessTask()
        self.attempts = 0
        self.rejected = []          # (reason, sequence) of every attempt that did not pass: "sampling" | "forcing" | "no_note"

    def execute(self, input_args: dict, max_rounds: Optional[int] = None, uniform_seed: int = 0) -> List[List[int]]:
        """input_args: the reference's input dictionary (generate.py:17-44).  Runs rounds of parallel decoding until
        `num_generate` sequences passed both validators (the reference retries one sequence at a time forever;
        `max_rounds` bounds it)."""
        self.model.eval()
        self.model.same_length = True                                   # model_initializer.py:49-50
        self.model.reset_length(1, self.memory_length)
        encoded_meta = self.preprocess_task.execute(dict(input_args))
        data = self.preprocess_task.input_data
        checker = InferenceTask(self.device)                            # only its validator is used here
        gen = BatchedGenerator(self.model, self.device, self.generation_length, self.memory_length)

        def accept(seq, teacher) -> bool:
            self.attempts += 1
            if seq is None:
                self.rejected.append(("sampling", None))
                return False
            try:
                teacher.validate_teacher_forced_sequence(seq)
            except Exception:
                self.rejected.append(("forcing", seq))
                return False
            if checker.validate_generated_sequence(seq):
                return True
            self.rejected.append(("no_note", seq))
            return False
        out, _ = gen.generate_stream(encoded_meta, data, data.temperature, data.top_k, data.num_generate, accept,
                                     top_p=getattr(data, "top_p", 1.0), seed=uniform_seed,
                                     max_attempts=None if max_rounds is None else max_rounds * data.num_generate)
        return out
