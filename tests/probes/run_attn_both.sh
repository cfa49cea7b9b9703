#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_dropout_gpu.py -m gpu -x -q -k "attn or dropout or mask or relattn" > gpurun_out/gpu_tests_attn.log 2>&1; echo "tests rc=$?" >> gpurun_out/gpu_tests_attn.log
tail -5 gpurun_out/gpu_tests_attn.log
AB_DROP=0.1 bash tests/probes/run_attn_prof.sh 2>&1 | head -6
