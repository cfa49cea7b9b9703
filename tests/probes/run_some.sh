cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
timeout 1200 python -m pytest tests -m gpu -q -x -k "embed or g1_ or configs or graph_step or dropout" 2>&1 | tail -6
python tests/probes/iter_probe.py 2>&1 | tail -8
