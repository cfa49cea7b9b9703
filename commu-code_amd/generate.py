#!/usr/bin/env python3
"""Generation driver with the reference's command line (generate.py:8-50):

    python commu-code_amd/generate.py --checkpoint_dir <checkpoint.pt> --output_dir <dir> --bpm 70 \\
        --audio_key aminor --time_signature 4/4 --pitch_range mid_high --num_measures 8 --inst acoustic_piano \\
        --genre newage --min_velocity 60 --max_velocity 80 --track_role main_melody --rhythm standard \\
        --chord_progression Am-Am-...-E --num_generate 3

checkpoint -> model (ModelInitializeTask), arguments -> meta tokens + chord components (PreprocessTask),
`num_generate` sequences decoded in parallel with chord forcing (InferenceTask), both validators.  The token
sequences are written to <output_dir>/sequences.json: turning them into MIDI files (sequence_postprocessor.py,
miditoolkit) is outside the hot path this package replaces -- the lists are exactly what
`PostprocessTask.execute(sequences=...)` takes.
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def parse_args():
    from commu_amd.midi_generator import meta
    model_arg_parser = argparse.ArgumentParser(description="Model Arguments")
    input_arg_parser = argparse.ArgumentParser(description="Input Arguments")
    model_arg_parser.add_argument("--checkpoint_dir", type=str)
    input_arg_parser.add_argument("--output_dir", type=str, required=True)
    input_arg_parser.add_argument("--bpm", type=int)
    input_arg_parser.add_argument("--audio_key", type=str, choices=list(meta.KEY_MAP.keys()))
    input_arg_parser.add_argument("--time_signature", type=str, choices=list(meta.TIME_SIG_MAP.keys()))
    input_arg_parser.add_argument("--pitch_range", type=str, choices=list(meta.PITCH_RANGE_MAP.keys()))
    input_arg_parser.add_argument("--num_measures", type=float)
    input_arg_parser.add_argument("--inst", type=str, choices=list(meta.INST_MAP.keys()))
    input_arg_parser.add_argument("--genre", type=str, default="cinematic", choices=list(meta.GENRE_MAP.keys()))
    input_arg_parser.add_argument("--track_role", type=str, choices=list(meta.TRACK_ROLE_MAP.keys()))
    input_arg_parser.add_argument("--rhythm", type=str, default="standard", choices=list(meta.RHYTHM_MAP.keys()))
    input_arg_parser.add_argument("--min_velocity", type=int, choices=range(1, 128))
    input_arg_parser.add_argument("--max_velocity", type=int, choices=range(1, 128))
    input_arg_parser.add_argument("--chord_progression", type=str, help="Chord progression ex) C-C-E-E-G-G ...")
    input_arg_parser.add_argument("--num_generate", type=int)
    input_arg_parser.add_argument("--top_k", type=int, default=32)
    input_arg_parser.add_argument("--temperature", type=float, default=0.95)
    # not in the reference (which retries rejected sequences forever, midi_inferrer.py:342-353): bound the retries
    input_arg_parser.add_argument("--max_rounds", type=int, default=None)
    return {"model_args": model_arg_parser, "input_args": input_arg_parser}


def main(model_args, input_args, training_cfg=None):
    import torch
    from commu_amd.midi_generator.meta import PreprocessTask
    from commu_amd.midi_generator.midi_inferrer import InferenceTask
    from commu_amd.midi_generator.model_initializer import ModelInitializeTask
    device = torch.device("cuda")
    init = ModelInitializeTask(model_args, map_location="cpu", device=device, training_cfg=training_cfg)
    model = init.execute()
    pre = PreprocessTask()
    in_args = dict(vars(input_args))
    max_rounds = in_args.pop("max_rounds", None)
    encoded_meta = pre.execute(in_args)
    task = InferenceTask(device)
    task(model=model, input_data=pre.input_data, inference_cfg=init.inference_cfg)
    sequences = task.execute(encoded_meta, max_rounds=max_rounds)
    os.makedirs(input_args.output_dir, exist_ok=True)
    with open(os.path.join(input_args.output_dir, "sequences.json"), "w") as f:
        json.dump({"encoded_meta": encoded_meta, "sequences": sequences}, f)
    return sequences


if __name__ == "__main__":
    margs, _ = parse_args()["model_args"].parse_known_args()
    iargs, _ = parse_args()["input_args"].parse_known_args()
    main(margs, iargs)
