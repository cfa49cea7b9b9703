#!/bin/bash
# A/B of two builds of the library on ONE box (box-to-box spread is +-3 %): lib/libcommu_hip_prev.so against the current
# lib/libcommu_hip.so, interleaved bench runs.  Make the previous build with:
#   git stash; python commu-code_amd/build.py; cp commu-code_amd/lib/libcommu_hip.so commu-code_amd/lib/libcommu_hip_prev.so; git stash pop; python commu-code_amd/build.py
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
L=commu-code_amd/lib
cp $L/libcommu_hip.so $L/libcommu_hip_new.so
for i in 1 2 3; do
  for v in prev new; do
    cp $L/libcommu_hip_$v.so $L/libcommu_hip.so
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-decode --no-extra $AB_FLAGS 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$v', d['ms_per_step'], d['value'])"
  done
done
cp $L/libcommu_hip_new.so $L/libcommu_hip.so
