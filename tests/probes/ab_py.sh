#!/bin/bash
# A/B of two versions of ONE python source on one box: ab_py.sh <path> ; expects <path>.prev beside it (e.g. from `git show HEAD:<path> > <path>.prev`)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
f=$1
cp $f $f.new
for i in 1 2 3; do
  for v in prev new; do
    cp $f.$v $f
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-decode --no-extra $AB_FLAGS 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$v', d['ms_per_step'], d['value'])"
  done
done
cp $f.new $f; rm -f $f.new
