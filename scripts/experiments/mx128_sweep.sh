#!/bin/bash
# k_mxgemm | k_mxgemm256 (MF = 16) | its 128-row form (MF = 8) | the library's default rule, MX-FP4 and e4m3 weight operands.
# Usage: scripts/experiments/mx128_sweep.sh > profiles/rNN_mx128_sweep.txt
cd "$(dirname "$0")/../.."
S=""
for m in 256 512 768 1024 1536 2048 3072 4096; do
  for nk in 12288,4096 4096,4096 22016,4096 4096,11008; do S="$S;$m,$nk"; done
done
S="$S;2048,16384,4096;2048,8192,8192;2048,28672,8192"
OPS=fp4,e4m3 SHAPES="${S#;}" python scripts/experiments/mx256_ab.py time 2>&1 | grep "^M"
