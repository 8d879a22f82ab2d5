#!/bin/bash
# A/B of the wave-tile height of the fused GEMM (MSQ_GEMM_MF=8: 128 x 64 wave tiles, two blocks per CU; 16: 256 x 64, AGPR accumulators)
b() { MSQ_GEMM_MF=$1 python bench.py --no-cpu-baseline --steps 300 "${@:3}" 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$2 MF=$1', round(j['value'],1), 'TF', round(j['roofline']['kernel_ms']*1000,1), 'us')"; }
for mf in 8 16 8 16; do b $mf posit; done
for mf in 8 16; do b $mf fp8 --outlier fp8_e4m3; done
for mf in 8 16; do b $mf posit_M8192 --M 8192 --steps 100; done
for mf in 8 16; do b $mf posit_M512 --M 512; done
