#!/bin/bash
# Ablation builds of k_qgemm3 (bf16-activation fused GEMM): libmsq_hip_abl<V>.so with -DMSQ_ABL=<V>
# (1 no LDS fragment reads, 2 no converts, 4 no packed-weight loads, 8 no activation staging, 16 no output stores; results are
# wrong by construction, timing only).  Usage: scripts/experiments/build_abl.sh 1 2 4 8 16 31; then scripts/experiments/abl_bench.py
set -e
cd "$(dirname "$0")/../../microscopiq-llm-quantization_amd/csrc"
OUT=../../scripts/experiments/abl; mkdir -p $OUT
for v in "$@"; do
  ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -DMSQ_ABL=$v $MSQ_ABL_EXTRA -c msq_gemm.hip -o $OUT/msq_gemm_a$v.o 2>/dev/null &&
    hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libmsq_hip_abl$v.so msq_quant.o msq_quant_lowp.o msq_quant_hw.o msq_pack_emit.o msq_pack_twopass.o msq_pack_unified.o msq_act.o msq_mx.o msq_kv.o msq_vec.o msq_gptq.o $OUT/msq_gemm_a$v.o &&
    rm $OUT/msq_gemm_a$v.o && echo built $v ) &
done
wait
