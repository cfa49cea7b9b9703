#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
echo "== gemm8 on"; timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-decode 2>/dev/null | tee gpurun_out/bench_g8on.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['time_share'])"
echo "== gemm8 off"; COMMU_GEMM8_OFF=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-decode 2>/dev/null | tee gpurun_out/bench_g8off.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['time_share'])"
