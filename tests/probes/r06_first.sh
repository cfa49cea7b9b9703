#!/bin/bash
# round 6, first call: the default bench line of the inherited tree, then two kernel traces of the bench step for the
# critical-path model (side streams on = the shipped schedule; off = every kernel alone on one queue: its isolated duration)
R=$GRAFT_REPO_ROOT
cd $R
python3 bench.py --no-cpu-baseline > gpurun_out/r06_base_bench.json 2> gpurun_out/r06_base_bench.err
tail -c 1500 gpurun_out/r06_base_bench.json | head -c 600; echo
cd /tmp; export TMPDIR=/tmp
for mode in side noside; do
  rm -rf /tmp/tr_$mode
  extra=""; [ $mode = noside ] && extra="--no-side-stream"
  rocprofv3 --kernel-trace -d /tmp/tr_$mode -o p -- python3 $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-decode --no-extra --no-graph $extra > /tmp/tr_$mode.log 2>&1
  grep -o '"ms_per_step": [0-9.]*' /tmp/tr_$mode.log | head -1
  python3 $R/tests/probes/trace_dump.py /tmp/tr_$mode/p_results.db $R/gpurun_out/r06_trace_$mode.csv
  gzip -f $R/gpurun_out/r06_trace_$mode.csv
done
ls -la $R/gpurun_out | tail -5
