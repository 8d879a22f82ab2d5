#!/bin/bash
# PMC passes for the fused GEMM (run on the GPU box through gpurun).  Usage: scripts/pmc_gpu.sh <tag> [bench args]
set -u
TAG=${1:-pmc}; shift || true
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_$TAG; mkdir -p $OUT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES SQ_INSTS_VMEM_WR" \
           "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum" ; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline "$@" > $OUT/p$i.json 2> $OUT/p$i.err
done
python3 scripts/summarize_pmc.py $OUT > $OUT/summary.txt 2>&1; cat $OUT/summary.txt
