#!/bin/bash
# round deliverables in one GPU call: kernel stats of the bench step, PMC traffic of the attention kernels, full bench line
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
tag=${1:-r02}
bash tests/probes/run_prof.sh $tag > /dev/null 2>&1
cp gpurun_out/kstats_$tag.txt gpurun_out/${tag}_kernel_stats.txt
bash tests/probes/pmc_traffic.sh > gpurun_out/pmc_traffic.log 2>&1
cp gpurun_out/pmc_traffic.json gpurun_out/${tag}_pmc_traffic.json
mkdir -p profiles; cp gpurun_out/${tag}_pmc_traffic.json profiles/      # so that the bench below finds the stamped traffic
python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
head -12 gpurun_out/${tag}_kernel_stats.txt | cut -c1-150
tail -2 gpurun_out/${tag}_kernel_stats.txt
python -c "
import json; d=json.load(open('gpurun_out/${tag}_bench.json')); print(d['value'], d['ms_per_step'], d['roofline'])
for k,v in d.get('extra_rows',{}).items(): print(k, v['value'], v['ms_per_step'])
print(d['decode']['short_memory']['tokens_per_s'], d['decode']['long_memory']['tokens_per_s'], d['cpu_baseline'])
"
