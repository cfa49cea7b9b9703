#!/bin/bash
# instruction and wait counters of the eight-phase NT GEMM at one large-K shape (65536 x 1536 x 4096)
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
cat > /tmp/g1.py <<PY
import sys, os, torch
sys.path.insert(0, os.path.join("$R", "commu-code_amd"))
from commu_amd import ops
M, N, K = 65536, 1536, 4096
A = torch.randn(M, K, device="cuda").bfloat16(); W = torch.randn(N, K, device="cuda").bfloat16()
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
for _ in range(3): ops.gemm_nt(A, W, out=out)
torch.cuda.synchronize()
PY
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_SALU SQ_WAVES" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_LDS SQ_IFETCH SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_VALU_TRANS_F32 GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace -d /tmp/pg$i -o x -- python3 /tmp/g1.py >/dev/null 2>&1
  python3 $R/tests/probes/pmc_table.py /tmp/pg$i/x_results.db gemm_nt8
done
