#!/bin/bash
# PMC passes over the forward attention kernel at the bench shape: pmc_fwd3.sh <kernel-substring> [gen] [drop]
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
K=${1:-relattn_fwd3}; export COMMU_ATTN_FWD_GEN=${2:-3}; export AB_DROP=${3:-0.0}; export AB_WHAT=fwd AB_REPS=3 AB_B=64
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_SALU SQ_WAVES" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_LDS SQ_IFETCH SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_VALU_TRANS_F32 GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace -d /tmp/pq$i -o x -- python3 $R/tests/probes/attn_bench.py >/dev/null 2>&1
  python3 $R/tests/probes/pmc_table.py /tmp/pq$i/x_results.db $K
done
