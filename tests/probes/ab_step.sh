#!/bin/bash
# interleaved step-time A/B of two trees on one box: ab_step.sh <treeA> <treeB> [reps]   (trees relative to the repo root, "." = this tree)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
A=$1; B=$2; reps=${3:-3}
for r in $(seq $reps); do
  for t in $A $B; do
    echo -n "$t: "; (cd $t && python bench.py --no-extra --no-decode --no-cpu-baseline --no-graph ${AB_ARGS} | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])")
  done
done
