cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
timeout 1500 python -m pytest tests/test_decode_gpu.py tests/test_kernels_gpu.py -m gpu -q -x -k "decode or layer_tail" 2>&1 | tail -4
bash tests/probes/run_decode_prof.sh 2>&1 | grep -E "==|decode_attn|decode_tail|sample|forcing"
timeout 600 python tests/probes/tail_ab.py 2>&1 | grep "tail 1"
