#!/bin/bash
# Variant builds of ONE object of libmsq_hip.so: scripts/experiments/build_q256.sh [-s <stem>] <name> "<extra hipcc flags>" [<name> "<flags>" ...]
# stem = msq_gemm256 (default) | msq_mxgemm256 | msq_gemm | ... -> scripts/experiments/abl/libmsq_hip_<stem tail>_<name>.so; every other object
# comes from the product build.  Load with ab.py / mx_ab.py (several libraries interleaved in one process) or through MSQ_LIB_OVERRIDE.
set -e
STEM=msq_gemm256
if [ "$1" = "-s" ]; then STEM=$2; shift 2; fi
cd "$(dirname "$0")/../../microscopiq-llm-quantization_amd/csrc"
OUT=../../scripts/experiments/abl; mkdir -p $OUT
TAG=${STEM#msq_}; TAG=${TAG/gemm256/q256}
OBJS=""
for o in msq_quant msq_quant_lowp msq_quant_hw msq_pack_emit msq_pack_twopass msq_pack_unified msq_act msq_mx msq_kv msq_vec msq_gptq msq_gemm msq_gemm256 msq_gemm256p msq_mxgemm256 msq_gemm_stream; do
  [ "$o" = "$STEM" ] || OBJS="$OBJS $o.o"
done
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result $flags -c $STEM.hip -o $OUT/${STEM}_$name.o 2>/dev/null &&
    hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libmsq_hip_${TAG}_$name.so $OBJS $OUT/${STEM}_$name.o &&
    rm $OUT/${STEM}_$name.o && echo built ${TAG}_$name ) &
done
wait
