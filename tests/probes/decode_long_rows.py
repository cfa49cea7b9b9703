"""to_completion rate of bench.py's 64-sequence generation against ForcedDecoder.LONG_ROWS / LONG_SPLITS (split-key graph policy)."""
import os, sys, json, argparse
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "commu-code_amd"))
import torch
import bench
from commu_amd.generate import ForcedDecoder
args = argparse.Namespace(layers=6, heads=8, d_model=512, d_inner=1024)
for rows, splits in ((0, 8), (16, 8), (24, 8), (32, 8), (48, 8), (32, 4), (24, 4), (64, 4)):
    ForcedDecoder.LONG_ROWS, ForcedDecoder.LONG_SPLITS = rows, splits
    import types
    # only the first part of decode_to_completion (the 64-sequence run)
    r = bench.decode_to_completion(torch.device("cuda"), args)
    print(rows, splits, r["tokens_per_s"], r["wall_ms"], r["bulk_256_sequences"]["continuous_slots"]["tokens_per_s"], flush=True)
