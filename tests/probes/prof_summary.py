"""Summarise a rocprofv3 rocpd sqlite result: per-kernel time (top_kernels) and PMC counter sums."""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
cur = con.cursor()
try:
    rows = list(cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
    print("# calls total_us avg_us pct name")
    for n, c, t, a, p in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 25]:
        print(f"{c:6d} {t:12.1f} {a:10.2f} {p:6.2f}  {n[:110]}")
except Exception as e:
    print("no top_kernels:", e)
try:
    q = ("select k.kernel_name, p.counter_name, sum(p.value), count(*) from counters_collection p "
         "join kernels k on k.dispatch_id = p.dispatch_id group by 1,2")
    cols = [d[1] for d in cur.execute("pragma table_info(counters_collection)")]
    print("# counters_collection columns:", cols)
    rows = list(cur.execute("select * from counters_collection limit 3"))
    for r in rows:
        print(r)
except Exception as e:
    print("no counters:", e)
