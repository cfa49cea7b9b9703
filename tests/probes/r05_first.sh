set -x
mkdir -p gpurun_out/r05a
python -m pytest tests/test_ddp_gpu.py tests/test_model_gpu.py -x -q -m gpu > gpurun_out/r05a/tests_ddp_model.log 2>&1; echo "rc=$?" >> gpurun_out/r05a/tests_ddp_model.log
python bench.py --no-extra --no-decode > gpurun_out/r05a/bench_short.json 2> gpurun_out/r05a/bench_short.err
bash tests/probes/step_traffic.sh > gpurun_out/r05a/step_traffic.log 2>&1
cp gpurun_out/step_traffic.json gpurun_out/r05a/ 2>/dev/null
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d /tmp/ks -o x -- python3 $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-decode --no-extra > /tmp/ks.log 2>&1; ls /tmp/ks; cp /tmp/ks/x_kernel_stats.csv $GRAFT_REPO_ROOT/gpurun_out/r05a/kernel_stats.csv)
tail -5 gpurun_out/r05a/tests_ddp_model.log
