#!/bin/bash
# Variant builds of msq_gemm256.hip: scripts/experiments/build_q256.sh <name> "<extra hipcc flags>" [<name> "<flags>" ...]
# -> scripts/experiments/abl/libmsq_hip_q256_<name>.so (every other object comes from the product build); load with ab.py under
# MSQ_GEMM_256=1, or through MSQ_LIB_OVERRIDE.
set -e
cd "$(dirname "$0")/../../microscopiq-llm-quantization_amd/csrc"
OUT=../../scripts/experiments/abl; mkdir -p $OUT
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result $flags -c msq_gemm256.hip -o $OUT/msq_gemm256_$name.o 2>/dev/null &&
    hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libmsq_hip_q256_$name.so msq_quant.o msq_quant_lowp.o msq_quant_hw.o msq_pack_emit.o msq_pack_twopass.o msq_pack_unified.o msq_act.o msq_mx.o msq_kv.o msq_vec.o msq_gptq.o msq_gemm.o $OUT/msq_gemm256_$name.o &&
    rm $OUT/msq_gemm256_$name.o && echo built $name ) &
done
wait
