#!/bin/bash
# ablations of relattn_bwd_q3_kernel at the bench shape (COMMU_Q3_ABL bits: 1 stores, 2 dS ring, 4 dropout hash, 8 V / Rd loads, 16 exp)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
for abl in ${ABLS:-0 1 3 7 15 31 4 8}; do
  out=/tmp/prof_abl; rm -rf $out; mkdir -p $out
  COMMU_Q3_ABL=$abl COMMU_ATTN_KV_GEN=4 AB_DROP=0.1 AB_B=64 AB_REPS=4 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o run -- python3 tests/probes/attn_bench.py > $out/log.txt 2>&1
  f=$(find $out -name "*kernel_stats.csv" | head -1)
  python3 - "$f" $abl <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    if "bwd_q3" in r["Name"]:
        print(f"abl {sys.argv[2]:>3s}: {float(r['AverageNs'])/1e3:8.1f} us  {r['Name'][:70]}")
PY
done
