#!/bin/bash
# usage: tools/pmc_generic.sh <outtag> <script.py> [args...]   (on the GPU box)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=$1; shift
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SMEM SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_TRANS_F32 GRBM_GUI_ACTIVE" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_LDS_IDX_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  t=$(echo $set | cut -c1-12 | tr " " "_")
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/${TAG}_$t -- python3 "$@" > /dev/null 2>&1
done
