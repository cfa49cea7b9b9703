#!/bin/bash
# HBM traffic of the attention kernels at the default bench shape: two rocprofv3 PMC passes (FETCH_SIZE and
# WRITE_SIZE do not fit one pass), then per-launch bytes -> gpurun_out/pmc_traffic.json (copy to profiles/).
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d /tmp/pt_$c -o x -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-decode --no-extra > /tmp/pt_$c.log 2>&1
done
python3 $R/tests/probes/pmc_traffic.py /tmp/pt_FETCH_SIZE/x_results.db /tmp/pt_WRITE_SIZE/x_results.db $R/gpurun_out/pmc_traffic.json
