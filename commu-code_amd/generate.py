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
    # not in the reference (which is fp32 throughout, train.py:48): run generation in the reference's own arithmetic --
    # fp32 weights, activations, K/V cache and products -- instead of the bf16 throughput path.  The mode in which greedy
    # decoding (--temperature 0) reproduces the reference's tokens wherever the top-1 / top-2 logit gap exceeds fp32
    # summation-order noise (~1e-6 of the logit range; the bf16 path needs a gap above its ~1e-2 logit error).
    model_arg_parser.add_argument("--parity", action="store_true")
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
    # not in the reference (top-k only, generate.py:43): nucleus filter applied after top-k; 1.0 = off
    input_arg_parser.add_argument("--top_p", type=float, default=1.0)
    # not in the reference (which retries rejected sequences forever, midi_inferrer.py:342-353): bound the retries
    input_arg_parser.add_argument("--max_rounds", type=int, default=None)
    # not in the reference: replicas of the generator, one per GPU (default: every visible GPU, at most num_generate)
    input_arg_parser.add_argument("--gpus", type=int, default=None)
    return {"model_args": model_arg_parser, "input_args": input_arg_parser}


from commu_amd.midi_generator.replicas import generate_on_device, replica_worker, split_num_generate  # noqa: E402


def main(model_args, input_args, training_cfg=None, device_indices=None):
    """`num_generate` sequences are independent (the reference produces them one after the other,
    midi_inferrer.py:338-354): with several GPUs visible they are split across them as REPLICAS -- one process per
    GPU, each with its own model copy and its share, no collective -- and concatenated in
    device order.  --gpus 1 (or one visible device) keeps everything in this process.  `device_indices` (tests):
    explicit device of every replica, e.g. [0, 0] = two replica processes on the one GPU of a test box."""
    import torch
    in_args = dict(vars(input_args))
    max_rounds = in_args.pop("max_rounds", None)
    gpus = in_args.pop("gpus", None)
    if device_indices is None:
        n_vis = torch.cuda.device_count()
        device_indices = list(range(max(1, n_vis if gpus is None else min(gpus, n_vis))))
    shares = split_num_generate(in_args["num_generate"], len(device_indices))
    if len(shares) == 1:
        encoded_meta, sequences = generate_on_device(model_args, in_args, device_indices[0], shares[0], 0, max_rounds,
                                                     training_cfg)
    else:
        import torch.multiprocessing as mp
        ctx = mp.get_context("spawn")          # (never fork / exec a process that has initialised the GPU)
        q = ctx.Queue()
        procs = [ctx.Process(target=replica_worker, args=(r, device_indices[r], model_args, in_args, share, max_rounds,
                                                    training_cfg, q)) for r, share in enumerate(shares)]
        for p_ in procs:
            p_.start()
        # a replica that dies natively (GPU fault, OOM kill) never reports: poll the queue and the processes
        import queue as _queue
        results = {}
        while len(results) < len(procs):
            try:
                k_, v_ = q.get(timeout=1.0)
                results[k_] = v_
            except _queue.Empty:
                dead = [(r, p_.exitcode) for r, p_ in enumerate(procs)
                        if r not in results and not p_.is_alive() and p_.exitcode not in (None, 0)]
                if dead:
                    for p_ in procs:
                        if p_.is_alive():
                            p_.terminate()
                    raise RuntimeError(f"generation replica {dead[0][0]} died with exit code {dead[0][1]}")
        for p_ in procs:
            p_.join()
        bad = [v for v in results.values() if isinstance(v, str)]
        if bad:
            raise RuntimeError("generation replica failed:\n" + bad[0])
        encoded_meta = results[0][0]
        sequences = [seq for r in range(len(shares)) for seq in results[r][1]]
    os.makedirs(input_args.output_dir, exist_ok=True)
    with open(os.path.join(input_args.output_dir, "sequences.json"), "w") as f:
        json.dump({"encoded_meta": encoded_meta, "sequences": sequences}, f)
    return sequences


if __name__ == "__main__":
    margs, _ = parse_args()["model_args"].parse_known_args()
    iargs, _ = parse_args()["input_args"].parse_known_args()
    main(margs, iargs)
