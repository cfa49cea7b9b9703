#!/bin/bash
# decode deliverables: per-kernel profile (eager launches), layer-tail phase trace, loop-stage trace, decode rows of the bench
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
bash tests/probes/run_decode_prof.sh > gpurun_out/r03_decode_kernels.txt 2>&1; head -30 gpurun_out/r03_decode_kernels.txt | cut -c1-150
{ echo "# python tests/probes/tail_trace.py  (decode layer-tail launch, 128 workgroups x 20 launches, 100 MHz timestamps)"; python tests/probes/tail_trace.py 2>&1 | grep -v amdgpu.ids;
  echo; echo "# python tests/probes/loop_trace.py  (sampling step / post / pre launch, 64 sequences x 20 launches)"; python tests/probes/loop_trace.py 2>&1 | grep -v amdgpu.ids; } > gpurun_out/r03_decode_phase_trace.txt
cat gpurun_out/r03_decode_phase_trace.txt
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_decode_rows.json
import argparse, json, torch, bench
a = argparse.Namespace(layers=6, heads=8, d_model=512, d_inner=1024)
for klen in (11, 1000):
    print(json.dumps(bench.decode_bench(torch.device("cuda"), a, klen)))
PY
