#!/bin/bash
# isolated attention kernels (bench shape, B=64): rocprofv3 per-kernel times
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
out=gpurun_out/prof_attn
echo "AB_DROP=${AB_DROP:-0}"
rm -rf $out; mkdir -p $out
AB_B=64 AB_REPS=5 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o run -- python3 tests/probes/attn_bench.py > $out/log.txt 2>&1
f=$(find $out -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print(f"{int(r['Calls']):6d} {float(r['AverageNs'])/1e3:10.2f} us  {r['Name'][:100]}")
PY
tail -2 $out/log.txt
find $out -size +8M -delete
