#!/bin/bash
# Variant builds of msq_gemm.hip: scripts/experiments/build_variant.sh <name> "<extra hipcc flags>" [<name> "<flags>" ...]
# -> scripts/experiments/abl/libmsq_hip_<name>.so (the other objects come from the product build), for ab.py / MSQ_LIB_OVERRIDE.
set -e
cd "$(dirname "$0")/../../microscopiq-llm-quantization_amd/csrc"
OUT=../../scripts/experiments/abl; mkdir -p $OUT
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result $flags -c msq_gemm.hip -o $OUT/msq_gemm_$name.o 2>/dev/null &&
    hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libmsq_hip_$name.so msq_quant.o msq_quant_lowp.o msq_quant_hw.o msq_pack_emit.o msq_pack_twopass.o msq_pack_unified.o msq_act.o msq_mx.o msq_kv.o msq_vec.o msq_gptq.o $OUT/msq_gemm_$name.o &&
    rm $OUT/msq_gemm_$name.o && echo built $name ) &
done
wait
