#!/bin/bash
# A/B of the staggered forward attention kernel (COMMU_ATTN_FWD_STAG) at the bench shape, then its parity tests
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
for rep in 1 2 3; do
for s in 0 1 2; do
  echo "STAG=$s"; COMMU_ATTN_FWD_STAG=$s AB_B=64 AB_WHAT=fwd AB_REPS=20 AB_DROP=0.1 python tests/probes/attn_bench.py
done; done

