cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
PROF_STEPS=20 PROF_WARMUP=5 bash tests/probes/run_prof.sh b8_eager --batch-per-gpu 8 --no-side-stream > /dev/null 2>&1
PROF_STEPS=20 PROF_WARMUP=5 bash tests/probes/run_prof.sh b8_graph --batch-per-gpu 8 --graph > /dev/null 2>&1
head -14 gpurun_out/kstats_b8_eager.txt | cut -c1-140; tail -1 gpurun_out/kstats_b8_eager.txt
head -6 gpurun_out/kstats_b8_graph.txt | cut -c1-140; tail -1 gpurun_out/kstats_b8_graph.txt
