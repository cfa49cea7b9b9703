"""Print avg kernel durations (us) from a rocprofv3 sqlite db for kernels matching a substring."""
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
for n, c, t, a, p in con.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    if len(sys.argv) < 3 or sys.argv[2] in n:
        import re
        m = re.search(r"(\w+_kernel(<[^>]*>)?)", n)
        print(f"{(m.group(1) if m else n[:40]):45s} calls={c:4d} avg_us={a:10.2f}")

if len(sys.argv) > 3:
    for name, dur in con.execute("select k.kernel_name, (k.end - k.start) from kernels k"):
        if sys.argv[2] in name:
            print("   dispatch_us", dur / 1000.0)
