cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_configs_gpu.py -m gpu -q -x -k "graph_step" 2>&1 | tail -40
