"""Dump the kernel records of a rocprofv3 kernel trace (rocpd sqlite) as CSV: start_ns,end_ns,queue,name (relative to the first
kernel).  usage: trace_dump.py <results.db> <out.csv>"""
import csv
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
qcol = next((c for c in ("queue_id", "stream_id", "queue", "stream") if c in cols), None)
ncol = next((c for c in ("kernel_name", "name", "kernel") if c in cols), None)
rows = list(con.execute(f"select start, end, {qcol or 0}, {ncol} from kernels order by start"))
t0 = rows[0][0]
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["start_ns", "end_ns", "queue", "name"])
    for s, e, q, n in rows:
        w.writerow([s - t0, e - t0, q, n.replace("(anonymous namespace)::", "")[:96]])
print(len(rows), "kernels ->", sys.argv[2], "columns:", cols)
