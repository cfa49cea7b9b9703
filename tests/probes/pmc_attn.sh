#!/bin/bash
# usage: pmc_attn.sh <kernel-substring> <AB_WHAT>   -- several PMC passes over the attention bench
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
K=${1:-relattn_fwd}; export AB_WHAT=${2:-fwd}; export AB_REPS=3
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_SALU SQ_WAVES" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_LDS SQ_IFETCH SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU" \
           "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_INT32 GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace -d /tmp/pp$i -o x -- python3 $R/tests/probes/attn_bench.py >/dev/null 2>&1
  python3 $R/tests/probes/pmc_table.py /tmp/pp$i/x_results.db $K
done
