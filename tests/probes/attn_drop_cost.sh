#!/bin/bash
# what the dropout mask costs each attention kernel (isolated, bench shape): kernel trace with AB_DROP = 0 and 0.1
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
for d in ${ADC_DROPS:-0.0 0.1}; do
  rm -rf /tmp/adc_$d
  AB_B=64 AB_DROP=$d AB_REPS=6 rocprofv3 --kernel-trace --stats -d /tmp/adc_$d -o p -- python3 $R/tests/probes/attn_bench.py > /tmp/adc_$d.log 2>&1
  echo "== attention dropout $d"
  python3 $R/tests/probes/prof_summary.py /tmp/adc_$d/p_results.db 12 | grep -i "relattn\|band\|delta\|calls"
done
