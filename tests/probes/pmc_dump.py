"""Every dispatch of a rocprofv3 PMC run whose kernel name matches: grid, counters.  usage: pmc_dump.py db regex"""
import re, sqlite3, sys
con = sqlite3.connect(sys.argv[1]); pat = re.compile(sys.argv[2])
rows = {}
for did, name, grid, c, v in con.execute("select dispatch_id, kernel_name, grid_size, counter_name, sum(value) from counters_collection group by dispatch_id, counter_name order by dispatch_id"):
    if pat.search(name):
        rows.setdefault(did, [re.sub(r"\(anonymous namespace\)::|^void ", "", name)[:48], grid, {}])[2][c] = v
for did, (name, grid, cs) in rows.items():
    print(f"{did:5d} {name:48s} grid {grid:8d} " + "  ".join(f"{k}={v:.1f}" for k, v in sorted(cs.items())))
