cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
PROF_STEPS=6 PROF_WARMUP=14 bash tests/probes/run_prof.sh r03_default --no-graph --d-model 500 --heads 10 --d-inner 1000 --tgt-len 128 --mem-len 1024 --batch-per-gpu 256 --batch-chunk 4 > /dev/null 2>&1
cat gpurun_out/kstats_r03_default.txt | cut -c1-150 | head -36; tail -1 gpurun_out/kstats_r03_default.txt
