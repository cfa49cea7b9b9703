#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_ddp_gpu.py -x -q > gpurun_out/gpu_tests2.log 2>&1
echo "tests rc=$?" >> gpurun_out/gpu_tests2.log
tail -30 gpurun_out/gpu_tests2.log
