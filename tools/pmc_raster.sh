#!/bin/bash
# PMC passes over the raster-only driver (run on the GPU box via gpurun): one rocprofv3 run per counter set, each with
# --kernel-trace only (never combined with sys/hip traces).  Results: gpurun_out/pmc_<tag>/ ; aggregate with
# tools/pmc_aggregate.py.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
WL=${1:-north_star}
EXTRA=${2:-}            # e.g. train=1000: the state bench.py's trained_state times
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SMEM SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_TRANS_F32 GRBM_GUI_ACTIVE" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $set | cut -c1-12 | tr " " "_")
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmc_$tag -- python3 tools/raster_only.py $WL 8 $EXTRA > /dev/null 2>&1
done
# (with train=N the process also holds the training iterations: only the 8 measured passes -- the last dispatches -- are averaged)
LAST="$EXTRA"
case "$WL" in stage*) LAST=train=stage;; esac     # (a pipeline state: the Stage-I loop that builds it runs in the same process)
case "$LAST" in train=*) python3 tools/pmc_aggregate.py gpurun_out/pmc_summary.json 8;; *) python3 tools/pmc_aggregate.py gpurun_out/pmc_summary.json;; esac
