#!/bin/bash
# Ablation builds of k_mxgemm: libmsq_hip_mxabl<V>.so with -DMSQ_MXABL=<V> (1 no LDS fragment reads, 2 no weight
# loads, 4 no LDS-DMA staging, 8 no barrier, 16 no output stores; results are wrong by construction, timing only).
# Usage: scripts/experiments/build_mx_ablation.sh 1 2 4 8 16 31 ...   then   MSQ_LIB_OVERRIDE=<so> python scripts/experiments/mx_bench.py
set -e
cd "$(dirname "$0")/../../microscopiq-llm-quantization_amd/csrc"
OUT=../../scripts/experiments/abl; mkdir -p $OUT
for v in "$@"; do
  ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -DMSQ_MXABL=$v -c msq_gemm.hip -o $OUT/msq_gemm_$v.o 2>/dev/null &&
    hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libmsq_hip_mxabl$v.so msq_quant.o msq_quant_hw.o msq_pack_emit.o msq_pack_twopass.o msq_pack_unified.o msq_act.o msq_mx.o $OUT/msq_gemm_$v.o &&
    rm $OUT/msq_gemm_$v.o && echo built $v ) &
done
wait
