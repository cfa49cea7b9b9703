cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
timeout 1200 python -m pytest tests -m gpu -q -x -k "colsum or layernorm or g1_ or relattn_bwd or ddp or two_rank or overlapped" 2>&1 | tail -15
python tests/probes/graph_probe.py 2>&1 | head -3
