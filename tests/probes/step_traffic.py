"""Per-kernel HBM bytes of one optimiser step from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; KiB per dispatch).
gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE tallies 128-byte requests at 64 bytes -> x 2; WRITE_SIZE is
taken as reported and CHECKED here against a program that moves a known byte count (hbm_calib.py).
usage: step_traffic.py fetch.db write.db calib_fetch.db calib_write.db <steps in the run> out.json"""
import hashlib
import json
import os
import re
import sqlite3
import sys
from collections import defaultdict


def table(con, prefix):
    for (name,) in con.execute("select name from sqlite_master where type in ('table','view')"):
        if name == prefix or name.startswith(prefix):
            return name
    raise SystemExit(f"no table {prefix}* in the profile")


def symbol(name):
    """Demangled kernel name without namespace noise and without its argument list (template arguments kept)."""
    s_ = name.replace("(anonymous namespace)::", "")
    s_ = re.sub(r"^void ", "", s_)
    depth = 0
    for i, ch in enumerate(s_):
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            return s_[:i]
    return s_


def per_kernel(db, counter):
    """{kernel symbol: (sum of counter over its dispatches, dispatch count)}"""
    con = sqlite3.connect(db)
    acc, ids = defaultdict(float), defaultdict(set)
    for name, cname, val, did in con.execute(
            "select kernel_name, counter_name, value, dispatch_id from counters_collection"):
        if cname != counter:
            continue
        short = symbol(name)
        acc[short] += val
        ids[short].add(did)
    return {k: (acc[k], len(ids[k])) for k in acc}


fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
cfetch, cwrite = per_kernel(sys.argv[3], "FETCH_SIZE"), per_kernel(sys.argv[4], "WRITE_SIZE")
steps = int(sys.argv[5])
GIB = float(1 << 30)


def calib(tbl, pattern):
    hit = [(k, v) for k, v in tbl.items() if re.search(pattern, k)]
    if not hit:
        return None
    k, (kib, n) = max(hit, key=lambda x: x[1][0])
    return {"kernel": k[:80], "launches": n, "kib_per_launch_raw": kib / n}


cal = {"copy_1GiB_fetch": calib(cfetch, r"copy|Copy|elementwise"), "copy_1GiB_write": calib(cwrite, r"copy|Copy|elementwise"),
       "fill_1GiB_write": calib(cwrite, r"[Ff]ill")}
for k, v in cal.items():
    if v is not None:
        corr = 2.0 if k.endswith("fetch") else 1.0
        v["bytes_per_launch_corrected"] = v["kib_per_launch_raw"] * 1024.0 * corr
        v["ratio_to_1GiB"] = v["bytes_per_launch_corrected"] / GIB

rows = {}
for k in set(fetch) | set(write):
    fk, fn = fetch.get(k, (0.0, 0))
    wk, wn = write.get(k, (0.0, 0))
    rd, wr = 2.0 * fk * 1024.0 / steps, wk * 1024.0 / steps
    rows[k] = {"launches_per_step": round(max(fn, wn) / steps, 2), "read_bytes_per_step": rd, "write_bytes_per_step": wr,
               "bytes_per_step": rd + wr}
order = sorted(rows, key=lambda k: -rows[k]["bytes_per_step"])
total = sum(r["bytes_per_step"] for r in rows.values())
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "commu-code_amd"))
from commu_amd import source_stamp          # the same stamp bench.py computes: a stale profile is refused
out = {"shape": [6, 512, 8, 1024, 0, 64], "source_sha256": source_stamp.kernel_source_hash(), "switches": source_stamp.traffic_switches(), "steps_in_run": steps,
       "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) on `python3 bench.py --steps 4 "
                 "--warmup 2 --no-cpu-baseline --no-decode --no-extra`; bytes = 2 x FETCH_SIZE KiB + WRITE_SIZE KiB, summed over every "
                 "dispatch of the run and divided by its optimiser steps (warm-up, timed and event-sampled steps; the first step's one-off initialisation included)",
       "calibration": cal, "step_hbm_bytes": total,
       "step_read_bytes": sum(r["read_bytes_per_step"] for r in rows.values()),
       "step_write_bytes": sum(r["write_bytes_per_step"] for r in rows.values()),
       "kernels": {k: rows[k] for k in order}}
json.dump(out, open(sys.argv[6], "w"), indent=1)
print(json.dumps({k: out[k] for k in ("calibration", "step_hbm_bytes", "step_read_bytes", "step_write_bytes")}, indent=1))
for k in order[:25]:
    r = rows[k]
    print(f"{r['bytes_per_step'] / 1e9:8.3f} GB/step  rd {r['read_bytes_per_step'] / 1e9:7.3f}  wr {r['write_bytes_per_step'] / 1e9:7.3f}  x{r['launches_per_step']:6.1f}  {k[:90]}")
