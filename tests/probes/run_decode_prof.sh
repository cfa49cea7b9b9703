#!/bin/bash
# per-kernel times of the decode iteration (eager launches so that kernels are named), klen 1000 and 11
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
for klen in 1000 11; do
  out=/tmp/dprof_$klen; rm -rf $out; mkdir -p $out
  DP_KLEN=$klen rocprofv3 --kernel-trace --stats --output-format csv -d $out -o run -- python3 tests/probes/decode_prof.py > $out/log.txt 2>&1
  f=$(find $out -name "*kernel_stats.csv" | head -1)
  echo "== klen $klen"; tail -1 $out/log.txt | cut -c1-200
  python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:16]:
    print(f"{int(r['Calls']):6d} {float(r['AverageNs'])/1e3:9.2f} us {100*float(r['TotalDurationNs'])/tot:6.1f}%  {r['Name'][:100]}")
PY
done
