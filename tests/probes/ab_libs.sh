#!/bin/bash
# isolated attention kernels (B = 64, dropout 0.1) for a list of library builds lib/libcommu_hip_<tag>.so (AB_LIBS="prev new ...";
# "new" = the current library): rocprofv3 kernel averages, two interleaved rounds
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
L=commu-code_amd/lib
cp $L/libcommu_hip.so $L/libcommu_hip_new.so
for round in 1 2; do for v in ${AB_LIBS:-prev new}; do
  cp $L/libcommu_hip_$v.so $L/libcommu_hip.so
  rm -rf /tmp/ab_attn; AB_B=64 AB_DROP=0.1 AB_REPS=6 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ab_attn -o run -- python3 tests/probes/attn_bench.py > /dev/null 2>&1
  f=$(find /tmp/ab_attn -name "*kernel_stats.csv" | head -1)
  python3 - "$f" "$v" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
out = []
for r in rows:
    n = r['Name']
    for key in ("relattn_bwd_q", "relattn_bwd_kv2", "relattn_fwd3", "band_bwd"):
        if key in n: out.append(f"{key} {float(r['AverageNs'])/1e3:.1f}")
print(sys.argv[2], " | ".join(sorted(out)))
PY
done; done
cp $L/libcommu_hip_new.so $L/libcommu_hip.so
