#!/bin/bash
# one GPU call: full GPU test suite, then the bench (all rows), logs under gpurun_out/
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/gpu_tests.log
tail -15 gpurun_out/gpu_tests.log
timeout 900 python bench.py "$@" > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err; echo "bench rc=$?"
tail -5 gpurun_out/bench_full.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_full.json"))
print(json.dumps({k: d[k] for k in ("value", "ms_per_step", "roofline")}, indent=None))
for k, v in d.get("extra_rows", {}).items():
    print(k, v["value"], v["ms_per_step"], v["step_mfma_frac"], v["roofline"]["kernel"], v["roofline"]["frac"])
print(json.dumps(d.get("decode", {}))[:1500])
print(d.get("cpu_baseline"))
PY
