#!/bin/bash
# round-3 deliverables in one GPU call: kernel stats of the bench step, PMC traffic of the attention kernels, attention PMC table,
# decode kernel profile, full bench line
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
bash tests/probes/run_profiles.sh r03 > gpurun_out/run_profiles_r03.log 2>&1
tail -12 gpurun_out/run_profiles_r03.log
bash tests/probes/pmc_all_attn.sh > gpurun_out/pmc_all_attn.log 2>&1; tail -3 gpurun_out/pmc_all_attn.log
bash tests/probes/run_decode_prof.sh > gpurun_out/r03_decode_kernels.txt 2>&1; head -5 gpurun_out/r03_decode_kernels.txt
