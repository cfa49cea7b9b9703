"""Per-kernel PMC sums from a rocprofv3 rocpd sqlite file: mean per dispatch of each counter."""
import re
import sqlite3
import sys
from collections import defaultdict

con = sqlite3.connect(sys.argv[1])
cur = con.cursor()
filt = sys.argv[2] if len(sys.argv) > 2 else "relattn"
acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(set)
meta = {}
for name, cname, val, did, vg, ag, lds, dur in cur.execute(
        "select kernel_name, counter_name, value, dispatch_id, vgpr_count, accum_vgpr_count, lds_block_size, duration "
        "from counters_collection"):
    if filt not in name:
        continue
    m = re.search(r"(\w+_kernel)", name); short = m.group(1) if m else name[:40]
    acc[short][cname] += val
    cnt[short].add(did)
    meta[short] = (vg, ag, lds)
for k in acc:
    n = len(cnt[k])
    print(f"== {k}  dispatches={n} vgpr={meta[k][0]} agpr={meta[k][1]} lds={meta[k][2]}")
    for c, v in sorted(acc[k].items()):
        print(f"   {c:28s} {v / n:16.1f}")
