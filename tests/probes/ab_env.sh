#!/bin/bash
# the bench step with an environment variable at two values, three interleaved pairs: ab_env.sh NAME A B [bench args]
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
name=$1; a=$2; b=$3; shift 3
for i in 1 2 3; do for v in $a $b; do
  env $name=$v python3 bench.py --no-extra --no-decode --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name=$v', d['ms_per_step'])"
done; done
