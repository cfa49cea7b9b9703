cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
for t in $1 $2 $1 $2; do
  echo -n "$t: "; (cd $t && AB_B=64 AB_WHAT=fwd,bwd AB_REPS=20 AB_DROP=0.1 python tests/probes/attn_bench.py)
done
