#!/bin/bash
# PMC tables of the four attention kernels at the bench shape (B = 64, dropout 0.1) -> gpurun_out/r05_attention_pmc.txt
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export AB_DROP=0.1 AB_B=64
out=$PWD/gpurun_out/r05_attention_pmc.txt
{
  echo "# rocprofv3 --pmc <set> --kernel-trace -- python3 tests/probes/attn_bench.py  (AB_B=64 AB_DROP=0.1; one pass per counter set;"
  echo "# values summed over the device per dispatch; SQ_*CYCLES / SQ_ACTIVE_* / SQ_WAIT_* count quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES cycles)"
  for k in relattn_fwd3 relattn_bwd_q relattn_bwd_kv2 band_bwd; do
    bash tests/probes/pmc_attn.sh $k fwd,bwd 2>&1 | grep -v "^$"
  done
} > $out
wc -l $out; head -12 $out
