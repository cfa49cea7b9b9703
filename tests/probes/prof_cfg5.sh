#!/bin/bash
# kernel stats of the wide configuration (L12 D1024 H16 DI2048 T2048 M2048, 8 sequences)
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/c5
rocprofv3 --kernel-trace --stats -d /tmp/c5 -o p -- python3 $R/bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-decode --no-extra --layers 12 --d-model 1024 --heads 16 --d-inner 2048 --tgt-len 2048 --mem-len 2048 --batch-per-gpu 8 > /tmp/c5.log 2>&1
grep -o '"ms_per_step": [0-9.]*' /tmp/c5.log | head -1
python3 $R/tests/probes/prof_summary.py /tmp/c5/p_results.db 24 | cut -c1-150
