#!/bin/bash
# grouped weight-gradient launch: operand bytes (FETCH_SIZE) and the bench step, lib/libcommu_hip_prev.so against the current library
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
L=commu-code_amd/lib
cp $L/libcommu_hip.so $L/libcommu_hip_new.so
for v in prev new; do
  cp $L/libcommu_hip_$v.so $L/libcommu_hip.so
  rm -rf /tmp/tr_$v; (cd /tmp; rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/tr_$v -o x -- python3 $GRAFT_REPO_ROOT/tests/probes/tn8_reads.py > /dev/null 2>&1)
  echo "== $v"; python3 tests/probes/pmc_dump.py /tmp/tr_$v/x_results.db tn8
done
cp $L/libcommu_hip_new.so $L/libcommu_hip.so
bash tests/probes/ab_step.sh
