#!/bin/bash
# Operand re-reads of the NT GEMMs against a start-up skew between the workgroups that share an activation row block
# (COMMU_GEMM8_SKEW cycles per phase group): FETCH_SIZE per launch + isolated times
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for s in 0 1500 4000 10000; do
  export COMMU_GEMM8_SKEW=$s
  echo "== skew $s"
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/sk_$s -o x -- python3 $R/tests/probes/write_amp.py > /tmp/sk.log 2>&1
  python3 $R/tests/probes/pmc_dump.py /tmp/sk_$s/x_results.db gemm
  python3 $R/tests/probes/gemm_epi_bench.py 2>&1 | sed 's/| torch.*//' | grep -E "qkv|ff1|ff2|dhid"
done
