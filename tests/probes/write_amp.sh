#!/bin/bash
# Output-write amplification: WRITE_SIZE / FETCH_SIZE / request-size split of (a) the two store patterns of store_pattern.hip,
# (b) the step's NT GEMM shapes isolated.  -> gpurun_out/write_amp.txt
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/write_amp.txt; : > $O
for set in "WRITE_SIZE" "FETCH_SIZE" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_REQ_sum TCC_WRITE_sum TCC_WRITEBACK_sum TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-20)
  rocprofv3 --pmc $set --kernel-trace -d /tmp/wa_s_$tag -o x -- $R/tests/probes/bin/store_pattern > /tmp/wa_s.log 2>&1
  echo "== store_pattern: $set" >> $O
  python3 $R/tests/probes/pmc_dump.py /tmp/wa_s_$tag/x_results.db 'k<' | awk 'NR%10==5' >> $O
  rocprofv3 --pmc $set --kernel-trace -d /tmp/wa_g_$tag -o x -- python3 $R/tests/probes/write_amp.py > /tmp/wa_g.log 2>&1
  echo "== gemm shapes: $set" >> $O
  python3 $R/tests/probes/pmc_dump.py /tmp/wa_g_$tag/x_results.db 'gemm' >> $O
done
cat /tmp/wa_g.log >> $O
cat $O
