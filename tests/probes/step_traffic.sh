#!/bin/bash
# HBM byte budget of ONE optimiser step at the default bench shape: two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE
# do not share a pass) over `bench.py --steps 4 --warmup 2` (8 eager steps: 2 warm-up + 4 timed + the 2 event-sampled steps
# that follow the timed region since round 6), summed per kernel symbol, plus the same two
# passes over a calibration program that moves a known byte count (tests/probes/hbm_calib.py: a 1 GiB device copy and
# a 1 GiB fill) -> gpurun_out/step_traffic.json (copy to profiles/rNN_step_traffic.json).
# The counters are collected in their own runs (no --stats, no tracing domains beside --kernel-trace).
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d /tmp/st_$c -o x -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-decode --no-extra > /tmp/st_$c.log 2>&1
  rocprofv3 --pmc $c --kernel-trace -d /tmp/sc_$c -o x -- python3 $R/tests/probes/hbm_calib.py > /tmp/sc_$c.log 2>&1
done
python3 $R/tests/probes/step_traffic.py /tmp/st_FETCH_SIZE/x_results.db /tmp/st_WRITE_SIZE/x_results.db \
        /tmp/sc_FETCH_SIZE/x_results.db /tmp/sc_WRITE_SIZE/x_results.db 8 $R/gpurun_out/step_traffic.json
tail -3 /tmp/st_FETCH_SIZE.log
