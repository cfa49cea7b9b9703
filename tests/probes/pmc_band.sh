#!/bin/bash
# PMC passes for one kernel of the attention bench: pmc_band.sh <kernel-substring>
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
K=${1:-band_bwd}; export AB_B=64; export AB_REPS=2
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace -d /tmp/pb$i -o x -- python3 $R/tests/probes/attn_bench.py >/dev/null 2>&1
  python3 $R/tests/probes/pmc_table.py /tmp/pb$i/x_results.db $K
done
