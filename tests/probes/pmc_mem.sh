#!/bin/bash
# usage: pmc_mem.sh <kernel-substring> <AB_WHAT>  -- memory-path counters for one attention kernel
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
K=${1:-relattn_fwd}; export AB_WHAT=${2:-fwd}; export AB_REPS=3; export AB_B=${AB_B:-16}
i=0
for set in "TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_BUSY_avr"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace -d /tmp/pq$i -o x -- python3 $R/tests/probes/attn_bench.py >/dev/null 2>&1
  python3 $R/tests/probes/pmc_table.py /tmp/pq$i/x_results.db $K
done
