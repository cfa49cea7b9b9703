"""Stamp of the code a committed PMC profile was measured on: bench.py reports the HBM bytes of a profile under `profiles/`
only when this stamp matches the tree it runs from (a number measured on other kernels, or with other traffic-changing
switches, is refused).  Used by bench.py and by tests/probes/{pmc_traffic,step_traffic}.py -- one definition."""
import hashlib
import os

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(HERE), "csrc")


def kernel_source_hash():
    """sha256 over EVERY kernel source and header (csrc/*.hip, *.h: the memory-side kernels -- Adam, LayerNorm, posemb -- move
    bytes too) and the two host files that choose which kernels a step launches (ops.py, model/model.py)."""
    h = hashlib.sha256()
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h")))
    files += [os.path.join(HERE, "ops.py"), os.path.join(HERE, "model", "model.py")]
    for path in files:
        h.update(os.path.basename(path).encode())
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def traffic_switches():
    """The environment-driven switches of ops.py that change a step's HBM traffic without changing a source file."""
    from . import ops
    return {"FWD_SAVES_P": bool(ops.FWD_SAVES_P), "DELTA_KERNEL": bool(ops.DELTA_KERNEL), "STORE_ATTN_P": bool(ops.STORE_ATTN_P)}
