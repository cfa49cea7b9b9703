cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
timeout 1500 python -m pytest tests/test_decode_gpu.py tests/test_model_gpu.py -m gpu -q -x -s -k "${TAIL_K:-decode or generat or layer_tail}" 2>&1 | tail -15
timeout 600 python tests/probes/tail_ab.py 2>&1 | tail -12
