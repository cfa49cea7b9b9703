cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
timeout 1200 python -m pytest tests -m gpu -q -x -k "dropout or graph_step or relattn" 2>&1 | tail -6
