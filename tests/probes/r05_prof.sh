#!/bin/bash
# round-5 profile set: rocprofv3 kernel stats of the bench step, whole-step HBM traffic (two PMC passes + calibration), per-launch
# traffic of the attention kernels, the fp8 GEMM against the bf16 one, the descriptor probe
cd "$GRAFT_REPO_ROOT"
bash tests/probes/run_prof.sh r05 > /dev/null 2>&1
bash tests/probes/step_traffic.sh > gpurun_out/step_traffic.log 2>&1
bash tests/probes/pmc_traffic.sh > gpurun_out/pmc_traffic.log 2>&1
python3 tests/probes/fp8_bench.py > gpurun_out/fp8_vs_bf16.txt 2>&1
hipcc --offload-arch=gfx950 -O2 -o /tmp/srd_soffset tests/probes/srd_soffset.hip 2>/dev/null; /tmp/srd_soffset > gpurun_out/srd_soffset.txt 2>&1
tail -32 gpurun_out/step_traffic.log
head -36 gpurun_out/kstats_r05.txt
cat gpurun_out/fp8_vs_bf16.txt | grep -v amdgpu
