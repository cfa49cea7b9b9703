#!/bin/bash
# isolated attention kernels (B = 64, dropout 0.1) with the forward saving its probabilities (AB_SAVEP=1) against the backward
# recomputing them (AB_SAVEP=0): rocprofv3 kernel averages, interleaved
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
for v in 0 1 0 1; do
  rm -rf /tmp/ab_attn; AB_SAVEP=$v AB_B=64 AB_DROP=0.1 AB_REPS=6 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ab_attn -o run -- python3 tests/probes/attn_bench.py > /dev/null 2>&1
  f=$(find /tmp/ab_attn -name "*kernel_stats.csv" | head -1)
  python3 - "$f" "savep=$v" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
out, tot = [], 0.0
for r in rows:
    n = r['Name']
    for key in ("relattn_bwd_q", "relattn_bwd_kv2", "relattn_fwd3", "band_bwd", "attn_delta"):
        if key in n:
            out.append(f"{key} {float(r['AverageNs'])/1e3:.1f}"); tot += float(r['AverageNs'])/1e3
print(sys.argv[2], " | ".join(sorted(out)), f"| sum {tot:.1f}")
PY
done
