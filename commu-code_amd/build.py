#!/usr/bin/env python3
"""Build libcommu_hip.so (gfx950) in-tree: commu-code_amd/lib/libcommu_hip.so.

hipcc cross-compiles without a GPU.  Objects are cached by source mtime under build/.
Usage: python commu-code_amd/build.py [--force]
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OUT_DIR = os.path.join(HERE, "lib")
OBJ_DIR = os.path.join(HERE, "build")
LIB = os.path.join(OUT_DIR, "libcommu_hip.so")
ARCH = "gfx950"
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value",
         # keep MFMA accumulators in VGPRs (gfx950 has a unified file): no v_accvgpr_* copies around the VALU work
         "-mllvm", "-amdgpu-mfma-vgpr-form",
         "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def newest_header():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs.append(os.path.join(ROOT, "include", "commu_hip.h"))
    return max(os.path.getmtime(h) for h in hs)


def build(force=False, verbose=True):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OUT_DIR, exist_ok=True)
    os.makedirs(OBJ_DIR, exist_ok=True)
    hdr = newest_header()
    objs, procs = [], []
    for s in sources():
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ_DIR, s[:-4] + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr):
            if verbose:
                print("hipcc", s, flush=True)
            procs.append((s, subprocess.Popen([hipcc] + FLAGS + ["-c", src, "-o", obj])))
    for s, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {s}")
    if procs or not os.path.exists(LIB):
        subprocess.check_call([hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs)
        if verbose:
            print("linked", LIB, flush=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
