#!/bin/bash
# forward attention (generation 3) at the bench shape, lib/libcommu_hip_prev.so against the current library, interleaved
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
L=commu-code_amd/lib
cp $L/libcommu_hip.so $L/libcommu_hip_new.so
for v in prev new prev new prev new; do
  cp $L/libcommu_hip_$v.so $L/libcommu_hip.so
  for d in 0.0 0.1; do echo -n "$v "; COMMU_ATTN_FWD_GEN=3 python3 tests/probes/attn3_fwd.py $d; done
done
cp $L/libcommu_hip_new.so $L/libcommu_hip.so
