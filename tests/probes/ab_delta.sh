#!/bin/bash
# bwd_q launch time (HIP events inside the bench step) with delta computed in the kernel against the separate attn_delta launch
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
for i in 1 2; do for v in 1 0; do COMMU_DELTA_KERNEL=$v python - <<EOF2
import os,sys,json,io,contextlib
sys.path.insert(0,"commu-code_amd")
from commu_amd import ops
ops.DELTA_KERNEL = os.environ["COMMU_DELTA_KERNEL"]=="1"
sys.argv=["bench.py","--steps","20","--warmup","5","--no-cpu-baseline","--no-decode","--no-extra"]
buf=io.StringIO()
with contextlib.redirect_stdout(buf):
    import runpy; runpy.run_path("bench.py", run_name="__main__")
d=json.loads(buf.getvalue().strip().splitlines()[-1]); print("delta kernel" if ops.DELTA_KERNEL else "in bwd_q   ", d["ms_per_step"], d["roofline"]["avg_launch_ms"])
EOF2
done; done
