#!/bin/bash
# kernel trace of a few eager steps at the bench shape -> idle time between dependent launches per queue
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/sg
rocprofv3 --kernel-trace -d /tmp/sg -o p -- python3 $R/bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-decode --no-extra --no-graph > /tmp/sg.log 2>&1
grep -v simple_timer /tmp/sg.log | head -30 | cut -c1-300; ls /tmp/sg
python3 $R/tests/probes/stream_gaps.py /tmp/sg/p_results.db $SG_WINDOW
