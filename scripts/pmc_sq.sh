#!/bin/bash
# SQ-level PMC passes over scripts/time_kernels.py (run on the GPU box via gpurun); results under gpurun_out/sq_<tag>/
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-base}; shift
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVES" \
           "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/sq_$TAG/p$i -- python3 $R/scripts/time_kernels.py "$@" > $R/gpurun_out/sq_$TAG/p$i.log 2>&1
done
