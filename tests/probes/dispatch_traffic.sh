#!/bin/bash
# Per-dispatch HBM bytes of the GEMM launches of one optimiser step (see dispatch_traffic.py) -> gpurun_out/dispatch_traffic.txt
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d /tmp/dt_$c -o x -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-decode --no-extra ${BENCH_ARGS} > /tmp/dt_$c.log 2>&1
done
python3 $R/tests/probes/dispatch_traffic.py /tmp/dt_FETCH_SIZE/x_results.db /tmp/dt_WRITE_SIZE/x_results.db 3 "${DT_PAT:-gemm|layernorm|transpose}" $R/gpurun_out/${DT_OUT:-dispatch_traffic.txt} | tail -${DT_TAIL:-150}
