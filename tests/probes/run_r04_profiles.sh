#!/bin/bash
# round-4 deliverables in one GPU call: kernel stats of the bench step, PMC traffic of the attention kernels (stamped), the
# full bench line, PMC tables of the attention kernels (generation-3 forward with and without dropout beside the 16x16
# family), decode kernel profile
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
bash tests/probes/run_profiles.sh r04 > gpurun_out/run_profiles_r04.log 2>&1
tail -14 gpurun_out/run_profiles_r04.log
out=$PWD/gpurun_out/r04_attention_pmc.txt
{
  echo "# rocprofv3 --pmc <set> --kernel-trace -- python3 tests/probes/attn_bench.py  (B=64 H=8 T=1024 M=0; one pass per counter set;"
  echo "# values summed over the device per dispatch; SQ_*CYCLES / SQ_ACTIVE_* / SQ_WAIT_* count quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES cycles)"
  echo "## relattn_fwd3 (generation 3), no dropout"
  bash tests/probes/pmc_fwd3.sh relattn_fwd3 3 0.0 2>&1 | grep -v "^$"
  echo "## relattn_fwd3 (generation 3), dropout 0.1"
  bash tests/probes/pmc_fwd3.sh relattn_fwd3 3 0.1 2>&1 | grep -v "^$"
  export AB_DROP=0.1 AB_B=64
  for k in relattn_fwd2 relattn_bwd_q relattn_bwd_kv2 band_bwd; do
    echo "## $k (16x16 family, dropout 0.1)"
    bash tests/probes/pmc_attn.sh $k fwd,bwd 2>&1 | grep -v "^$"
  done
} > $out
wc -l $out
bash tests/probes/run_decode_prof.sh > gpurun_out/r04_decode_kernels.txt 2>&1; head -5 gpurun_out/r04_decode_kernels.txt
