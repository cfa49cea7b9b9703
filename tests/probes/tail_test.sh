cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
timeout 900 python -m pytest tests/test_decode_gpu.py -m gpu -q -x -s -k "layer_tail or long_memory" 2>&1 | tail -15
timeout 600 python tests/probes/tail_ab.py 2>&1 | tail -12
