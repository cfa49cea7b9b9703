"""Per-DISPATCH HBM bytes of the GEMM launches of the last optimiser step of a step_traffic.sh run (which of the shapes that
share a kernel symbol re-reads its operands).  The two PMC passes launch the same sequence, so dispatches are joined by
their ordinal.  usage: dispatch_traffic.py fetch.db write.db <steps in the run> <symbol regex> out.txt"""
import re
import sqlite3
import sys

sys.path.insert(0, __import__("os").path.dirname(__file__))


def symbol(name):
    s_ = re.sub(r"^void ", "", name.replace("(anonymous namespace)::", ""))
    depth = 0
    for i, ch in enumerate(s_):
        depth += ch == "<"
        depth -= ch == ">"
        if ch == "(" and depth == 0:
            return s_[:i]
    return s_


def dispatches(db, counter):
    con = sqlite3.connect(db)
    cols = [r[1] for r in con.execute("pragma table_info(counters_collection)")]
    grid = "grid_size_x" if "grid_size_x" in cols else ("grid_size" if "grid_size" in cols else None)
    q = f"select dispatch_id, kernel_name, {grid or '0'}, sum(value) from counters_collection where counter_name = ? group by dispatch_id order by dispatch_id"
    return [(d, symbol(n), g, v) for d, n, g, v in con.execute(q, (counter,))], cols


f, cols = dispatches(sys.argv[1], "FETCH_SIZE")
w, _ = dispatches(sys.argv[2], "WRITE_SIZE")
steps, pat = int(sys.argv[3]), re.compile(sys.argv[4])
assert len(f) == len(w), (len(f), len(w))
n = len(f) // steps
lines = [f"columns of counters_collection: {cols}", f"{len(f)} dispatches, last step = the last {n}"]
for k, ((_, s1, g, fv), (_, s2, _, wv)) in enumerate(zip(f[-n:], w[-n:])):
    if s1 != s2:
        lines.append(f"{k:5d} ORDER MISMATCH {s1[:40]} / {s2[:40]}")
        continue
    if pat.search(s1):
        lines.append(f"{k:5d} grid {g:7d}  rd {2 * fv * 1024 / 1e6:9.1f} MB  wr {wv * 1024 / 1e6:9.1f} MB  {s1[:70]}")
open(sys.argv[5], "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:400]))
