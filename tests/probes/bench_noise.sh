#!/bin/bash
# run-to-run spread of the bench step time, with and without the side stream
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
for flags in "" "--no-side-stream"; do
  echo "flags: $flags"
  for i in 1 2 3 4; do
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-decode --no-extra $flags 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(' ', d['ms_per_step'], d['value'])"
  done
done
