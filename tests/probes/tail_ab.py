"""Decode iteration time with the layer-tail launch on / off, same process (klen 11 and 1000, graph)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "commu-code_amd"))
import argparse, torch
import bench
import commu_amd.generate as G
a = argparse.Namespace(layers=6, heads=8, d_model=512, d_inner=1024)
steps = int(os.environ.get("TAIL_STEPS", 256))
for klen in (11, 1000):
    for rep in range(2):
        for tail in (True, False):
            G.USE_LAYER_TAIL = tail
            r = bench.decode_bench(torch.device("cuda"), a, klen, steps=steps, graph=True)
            print(f"klen {klen} tail {int(tail)}: {r['ms_per_step']} ms/iteration, {r['tokens_per_s']} tok/s, "
                  f"frac {r['roofline']['frac']}", flush=True)
