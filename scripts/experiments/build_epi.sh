#!/bin/bash
# The 256-row kernels with the LDS-transposed epilogue (-DMSQ_EPI_DIRECT=0) -> scripts/experiments/abl/libmsq_hip_epilds.so, for epi_ab.py
# (the product build uses the register-exchange epilogue, store_wave_tile_direct of csrc/msq_gemm_common.h).
set -e
cd "$(dirname "$0")/../../microscopiq-llm-quantization_amd/csrc"
OUT=../../scripts/experiments/abl; mkdir -p $OUT
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -DMSQ_EPI_DIRECT=0"
hipcc $F -c msq_gemm256.hip -o $OUT/q256_epilds.o 2>/dev/null &
hipcc $F -c msq_mxgemm256.hip -o $OUT/mx256_epilds.o 2>/dev/null &
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libmsq_hip_epilds.so msq_quant.o msq_quant_lowp.o msq_quant_hw.o msq_pack_emit.o msq_pack_twopass.o msq_pack_unified.o msq_act.o msq_mx.o msq_kv.o msq_vec.o msq_gptq.o msq_gemm.o msq_gemm256p.o $OUT/q256_epilds.o $OUT/mx256_epilds.o
rm $OUT/q256_epilds.o $OUT/mx256_epilds.o; echo built epilds
