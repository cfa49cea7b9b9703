#!/bin/bash
# isolated NT GEMMs with their epilogues + the bench step: lib/libcommu_hip_prev.so against the current library; then the output
# write bytes of the current library's GEMMs (WRITE_SIZE pass)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
L=commu-code_amd/lib
cp $L/libcommu_hip.so $L/libcommu_hip_new.so
for i in 1 2; do for v in prev new; do
  cp $L/libcommu_hip_$v.so $L/libcommu_hip.so
  echo "== $v"; python3 tests/probes/gemm_epi_bench.py 2>&1 | sed 's/| torch.*//' | grep -v "^$"
done; done
cp $L/libcommu_hip_new.so $L/libcommu_hip.so
bash tests/probes/ab_step.sh
cd /tmp
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/wa_n -o x -- python3 $GRAFT_REPO_ROOT/tests/probes/write_amp.py > /tmp/wa_n.log 2>&1
python3 $GRAFT_REPO_ROOT/tests/probes/pmc_dump.py /tmp/wa_n/x_results.db gemm
