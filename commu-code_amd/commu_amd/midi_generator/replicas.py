"""Generation replicas: `num_generate` independent sequences split across GPUs, one process per GPU, no collective
(the reference generates them one after the other on one device, midi_inferrer.py:338-354).
Lives in the package (not in the generate.py script) so that spawned replica processes can import the worker."""
from __future__ import annotations


def split_num_generate(num_generate, n_replicas):
    """Shares of `num_generate` for `n_replicas` independent replicas (first ones take the remainder)."""
    n_replicas = max(1, min(int(n_replicas), int(num_generate)))
    base, rem = divmod(int(num_generate), n_replicas)
    return [base + (1 if r < rem else 0) for r in range(n_replicas)]


def generate_on_device(model_args, in_args, device_index, num_generate, uniform_seed, max_rounds, training_cfg=None):
    """One replica: checkpoint -> model on cuda:<device_index>, `num_generate` validated sequences."""
    import copy

    import torch
    from commu_amd.midi_generator.meta import PreprocessTask
    from commu_amd.midi_generator.midi_inferrer import InferenceTask
    from commu_amd.midi_generator.model_initializer import ModelInitializeTask
    torch.cuda.set_device(device_index)
    device = torch.device("cuda", device_index)
    init = ModelInitializeTask(model_args, map_location="cpu", device=device, training_cfg=training_cfg)
    model = init.execute()
    pre = PreprocessTask()
    args = copy.deepcopy(in_args)
    args["num_generate"] = num_generate
    encoded_meta = pre.execute(args)
    task = InferenceTask(device)
    task.uniform_seed = uniform_seed
    task(model=model, input_data=pre.input_data, inference_cfg=init.inference_cfg)
    return encoded_meta, task.execute(encoded_meta, max_rounds=max_rounds)


def replica_worker(rank, device_index, model_args, in_args, share, max_rounds, training_cfg, q):
    try:
        # distinct variates per replica: 1_000_003 apart (a replica's rounds / sequences use seed + 7919 r + b)
        q.put((rank, generate_on_device(model_args, in_args, device_index, share, 1_000_003 * rank, max_rounds,
                                        training_cfg)))
    except Exception:
        import traceback
        q.put((rank, traceback.format_exc()))
