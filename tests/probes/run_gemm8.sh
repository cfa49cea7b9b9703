#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_decode_gpu.py tests/test_configs_gpu.py -x -q -k "layernorm_fused or decode or hipgraph or sampl or dh50 or cached or pipeline or config" > gpurun_out/dec_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/dec_tests.log
tail -4 gpurun_out/dec_tests.log
cat > /tmp/dbench.py <<'PY'
import sys, json, types, os
sys.argv = ["bench.py"]
sys.path.insert(0, os.getcwd())
import bench, torch
from commu_amd import ops
args = types.SimpleNamespace(layers=6, heads=8, d_model=512, d_inner=1024)
for fuse in (True, False):
    ops.FUSE_DECODE_LN = fuse
    for kl in (11, 1000):
        r = bench.decode_bench(torch.device("cuda"), args, kl)
        print("fuse", fuse, "klen", kl, r["tokens_per_s"], r["ms_per_step"], flush=True)
PY
timeout 600 python /tmp/dbench.py 2>&1 | grep fuse
