#!/bin/bash
# round-6 profile set on the final tree: the default bench line, rocprofv3 kernel stats of the bench step, the two kernel traces
# of the critical-path model (side streams on / off), whole-step HBM traffic (two PMC passes + calibration), per-launch traffic of
# the attention kernels, the isolated attention kernels old pair / 32x32 pair
cd "$GRAFT_REPO_ROOT"
python3 bench.py > gpurun_out/r06_bench.json 2> gpurun_out/r06_bench.err
bash tests/probes/run_prof.sh r06 > /dev/null 2>&1
R=$GRAFT_REPO_ROOT
( cd /tmp; export TMPDIR=/tmp
  for mode in side noside; do
    rm -rf /tmp/tr_$mode
    extra=""; [ $mode = noside ] && extra="--no-side-stream"
    rocprofv3 --kernel-trace -d /tmp/tr_$mode -o p -- python3 $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-decode --no-extra --no-graph $extra > /tmp/tr_$mode.log 2>&1
    python3 $R/tests/probes/trace_dump.py /tmp/tr_$mode/p_results.db $R/gpurun_out/r06_trace_$mode.csv > /dev/null
    gzip -f $R/gpurun_out/r06_trace_$mode.csv
  done )
python3 tests/probes/critical_path.py gpurun_out/r06_trace_side.csv.gz gpurun_out/r06_trace_noside.csv.gz gpurun_out/r06_critical_path.json > gpurun_out/r06_critical_path.txt 2>&1
bash tests/probes/step_traffic.sh > gpurun_out/step_traffic.log 2>&1
bash tests/probes/pmc_traffic.sh > gpurun_out/pmc_traffic.log 2>&1
bash tests/probes/ab_q3.sh > gpurun_out/r06_q3_vs_pair.txt 2>&1
head -30 gpurun_out/kstats_r06.txt
tail -5 gpurun_out/step_traffic.log
head -40 gpurun_out/r06_critical_path.txt
