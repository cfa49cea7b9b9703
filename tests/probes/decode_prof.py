"""Decode-step kernel profile: run under rocprofv3 --kernel-trace (no hipGraph so kernels are named)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "commu-code_amd"))
import argparse, torch
import bench
a = argparse.Namespace(layers=6, heads=8, d_model=512, d_inner=1024)
klen = int(os.environ.get("DP_KLEN", 1000))
graph = bool(int(os.environ.get("DP_GRAPH", 0)))
print(bench.decode_bench(torch.device("cuda"), a, klen, steps=32, graph=graph))
