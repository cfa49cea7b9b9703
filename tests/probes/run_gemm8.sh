#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "relattn_bwd" > gpurun_out/band_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/band_tests.log
tail -4 gpurun_out/band_tests.log
for abl in 0 2 4; do echo "abl $abl"; COMMU_BAND_ABL=$abl bash tests/probes/run_attn_prof.sh 2>&1 | grep "band_bwd\|bwd_q"; done
