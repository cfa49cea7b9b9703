#!/bin/bash
# rocprofv3 kernel statistics of the bench step: run_prof.sh <tag> [bench args...]
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
tag=$1; shift
export TMPDIR=/tmp
out=gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o run -- python3 bench.py --steps ${PROF_STEPS:-6} --warmup ${PROF_WARMUP:-2} --no-cpu-baseline --no-decode --no-extra "$@" > $out/bench.json 2> $out/err.log
f=$(find $out -name "*kernel_stats.csv" | head -1)
{
  echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps ${PROF_STEPS:-6} --warmup ${PROF_WARMUP:-2} --no-cpu-baseline --no-decode --no-extra $@"
  python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print("# calls total_us avg_us pct name")
tot = sum(float(r['TotalDurationNs']) for r in rows) / 1e3
print(f"# sum of all kernel durations: {tot:.1f} us over the whole run ({len(rows)} kernels)")
for r in rows[:32]:
    print(f"{int(r['Calls']):6d} {float(r['TotalDurationNs'])/1e3:12.1f} {float(r['AverageNs'])/1e3:10.2f} {float(r['Percentage']):6.2f}  {r['Name'][:120]}")
PY
  python -c "import json; d=json.load(open('$out/bench.json')); print('# ms_per_step', d['ms_per_step'], 'tokens/s', d['value'])"
} > gpurun_out/kstats_$tag.txt 2>&1
cat gpurun_out/kstats_$tag.txt
rm -rf $out/*/*.csv.bak; find $out -size +8M -delete
