#!/bin/bash
# isolated attention kernels at the bench shape (B = 64, dropout 0.1): the 16x16 backward pair (generation 2) against the
# 32x32 pair of relattn_q3.hip + relattn_kv3.hip (generation 4), interleaved
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
for rep in 1 2; do
for gen in 2 4; do
  out=/tmp/prof_q3_$gen; rm -rf $out; mkdir -p $out
  COMMU_ATTN_KV_GEN=$gen AB_DROP=${AB_DROP:-0.1} AB_B=64 AB_REPS=6 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o run -- python3 tests/probes/attn_bench.py > $out/log.txt 2>&1
  f=$(find $out -name "*kernel_stats.csv" | head -1)
  echo "generation $gen"
  python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:7]:
    print(f"{int(r['Calls']):6d} {float(r['AverageNs'])/1e3:10.2f} us  {r['Name'][:90]}")
PY
done
done
