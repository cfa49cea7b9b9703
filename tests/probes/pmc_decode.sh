#!/bin/bash
# HBM traffic of the decode-step kernels (eager launches, 64 sequences, klen 1000 and 11): two rocprofv3 PMC passes per
# length (FETCH_SIZE and WRITE_SIZE do not fit one pass) -> gpurun_out/r03_decode_pmc.txt
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r03_decode_pmc.txt
{
echo "# rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE --kernel-trace -- python3 tests/probes/decode_prof.py   (DP_KLEN = klen, 32 iterations + warm-up)"
echo "# bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) KiB  (gfx950: FETCH_SIZE counts 128-byte requests as 64, MI355X_MICROARCH.md HBM section)"
for klen in 1000 11; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pd_$c
    DP_KLEN=$klen rocprofv3 --pmc $c --kernel-trace -d /tmp/pd_$c -o x -- python3 $R/tests/probes/decode_prof.py > /tmp/pd_$c.log 2>&1
  done
  echo "== klen $klen"
  python3 - /tmp/pd_FETCH_SIZE/x_results.db /tmp/pd_WRITE_SIZE/x_results.db $klen <<'PY'
import re, sqlite3, sys
from collections import defaultdict
def per_launch(db, counter):
    con = sqlite3.connect(db)
    acc, ids = defaultdict(float), defaultdict(set)
    for name, cname, val, did in con.execute("select kernel_name, counter_name, value, dispatch_id from counters_collection"):
        if cname != counter:
            continue
        m = re.search(r"(decode_attn_kernel|decode_tail_kernel<[^>]*>|sample_post_pre_kernel)", name)
        if m:
            acc[m.group(1)] += val
            ids[m.group(1)].add(did)
    return {k: (acc[k] / max(1, len(ids[k])), len(ids[k])) for k in acc}
f, w = per_launch(sys.argv[1], "FETCH_SIZE"), per_launch(sys.argv[2], "WRITE_SIZE")
klen = int(sys.argv[3])
for k in sorted(f):
    fb, wb = 2.0 * f[k][0] * 1024, w.get(k, (0.0, 0))[0] * 1024
    note = ""
    if k == "decode_attn_kernel":
        alg = 64 * 8 * (klen + 16) * 64 * 2 * 2          # K and V rows of every (sequence, head), bf16 (mid-run length)
        note = f"   algorithmic K+V bytes ~ {alg / 1e6:.1f} MB"
    print(f"  {k:44s} launches {f[k][1]:4d}  read {fb / 1e6:8.2f} MB  written {wb / 1e6:7.3f} MB{note}")
PY
done
} > $out 2>&1
cat $out
