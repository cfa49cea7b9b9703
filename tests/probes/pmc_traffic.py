"""FETCH_SIZE / WRITE_SIZE (KiB) per launch of the attention kernels -> bytes per launch (JSON for bench.py).
gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE counts 128-byte requests at 64 bytes -> x2."""
import hashlib
import json
import os
import re
import sqlite3
import sys
from collections import defaultdict

KERNELS = {"relattn_fwd2_kernel": "commu_relattn_fwd", "relattn_fwd3_kernel": "commu_relattn_fwd", "relattn_fwd_kernel": "commu_relattn_fwd",
           "relattn_bwd_q_kernel": "commu_relattn_bwd_q", "relattn_bwd_kv2_kernel": "commu_relattn_bwd_kv",
           "relattn_bwd_kv_kernel": "commu_relattn_bwd_kv", "relattn_bwd_kv3_kernel": "commu_relattn_bwd_kv",
           "relattn_bwd_q3_kernel": "commu_relattn_bwd_q"}


def per_launch(db, counter):
    con = sqlite3.connect(db)
    acc, ids = defaultdict(float), defaultdict(set)
    for name, cname, val, did in con.execute(
            "select kernel_name, counter_name, value, dispatch_id from counters_collection"):
        if cname != counter:
            continue
        m = re.search(r"(relattn_\w+_kernel)", name)
        if m and m.group(1) in KERNELS:
            acc[m.group(1)] += val
            ids[m.group(1)].add(did)
    return {k: acc[k] / max(1, len(ids[k])) for k in acc}


fetch, write = per_launch(sys.argv[1], "FETCH_SIZE"), per_launch(sys.argv[2], "WRITE_SIZE")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "commu-code_amd"))
from commu_amd import source_stamp          # the same stamp bench.py computes: a stale profile is refused
out = {"shape": [6, 512, 8, 1024, 0, 64], "source_sha256": source_stamp.kernel_source_hash(), "switches": source_stamp.traffic_switches(), "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on "
       "`python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-decode --no-extra`; bytes = (2*FETCH_SIZE + WRITE_SIZE) KiB",
       "fetch_kib_raw": {KERNELS[k]: fetch[k] for k in fetch}, "write_kib_raw": {KERNELS[k]: write[k] for k in write},
       "bytes_per_launch": {KERNELS[k]: (2.0 * fetch.get(k, 0.0) + write.get(k, 0.0)) * 1024.0 for k in fetch}}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
