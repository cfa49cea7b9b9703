cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
GP_B=64 python tests/probes/graph_probe.py
GP_B=8 python tests/probes/graph_probe.py
