#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/gpu_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/gpu_tests.log
tail -4 gpurun_out/gpu_tests.log
timeout 600 python bench.py --steps 10 --warmup 3 2>/dev/null | tee gpurun_out/bench_full.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value']); print(json.dumps(d['decode'], indent=1)); print(d['cpu_baseline'])"
