#!/bin/bash
# The attention backward on head SUBSETS (the size a chunked schedule would launch): per-kernel averages for H = 8, 4, 2, 1 at
# B = 64, scaled to 8 heads -- does a producer/consumer distance below the 256-MB Infinity Cache pay for the smaller launches?
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for h in 8 4 2 1; do
  rm -rf /tmp/ah_$h
  AB_H=$h AB_B=64 AB_DROP=0.1 AB_REPS=6 AB_SAVEP=0 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ah_$h -o run -- python3 tests/probes/attn_bench.py > /dev/null 2>&1
  f=$(find /tmp/ah_$h -name "*kernel_stats.csv" | head -1)
  python3 - "$f" "$h" <<'PY'
import csv, sys
h = int(sys.argv[2]); out = []; tot = 0.0
for r in csv.DictReader(open(sys.argv[1])):
    for key in ("relattn_bwd_q", "relattn_bwd_kv2", "relattn_fwd3", "band_bwd", "attn_delta"):
        if key in r['Name']:
            us = float(r['AverageNs']) / 1e3
            out.append(f"{key} {us:.1f} (x{8 // h} = {us * 8 / h:.1f})")
            if key != "relattn_fwd3": tot += us * 8 / h
print(f"H={h}: " + " | ".join(sorted(out)) + f" | backward per layer {tot:.0f} us")
PY
done
