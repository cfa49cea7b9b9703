#!/bin/bash
# round-5 profile set: rocprofv3 kernel stats of the bench step + whole-step HBM traffic (two PMC passes + calibration)
cd "$GRAFT_REPO_ROOT"
bash tests/probes/run_prof.sh r05 > /dev/null 2>&1
bash tests/probes/step_traffic.sh > gpurun_out/step_traffic.log 2>&1
tail -30 gpurun_out/step_traffic.log
head -40 gpurun_out/kstats_r05.txt
