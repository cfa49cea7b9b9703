#!/bin/bash
# the bench step, lib/libcommu_hip_prev.so against the current library, three interleaved pairs
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
L=commu-code_amd/lib
cp $L/libcommu_hip.so $L/libcommu_hip_new.so
for i in 1 2 3; do for v in prev new; do
  cp $L/libcommu_hip_$v.so $L/libcommu_hip.so
  python3 bench.py --no-extra --no-decode --no-cpu-baseline ${AB_ARGS} 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['ms_per_step'], {k: round(v,4) for k,v in d['roofline']['time_share'].items() if 'attn' in k})"
done; done
cp $L/libcommu_hip_new.so $L/libcommu_hip.so
