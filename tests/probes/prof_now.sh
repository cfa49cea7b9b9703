cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
PROF_STEPS=8 PROF_WARMUP=3 bash tests/probes/run_prof.sh r03a --no-graph > /dev/null 2>&1
cat gpurun_out/kstats_r03a.txt | cut -c1-170
