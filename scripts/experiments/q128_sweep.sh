#!/bin/bash
# k_qgemm3 | k_qgemm256 (MF = 16) | its 128-row form (MF = 8) | the library's default rule, over M x the Llama-2-7B projections.
# Usage: scripts/experiments/q128_sweep.sh > profiles/rNN_q128_sweep.txt
cd "$(dirname "$0")/../.."
S=""
for m in 128 256 384 512 640 768 1024 1280 1536 2048 2560 3072 4096; do
  for nk in 12288,4096 4096,4096 22016,4096 4096,11008; do S="$S;$m,$nk"; done
done
S="$S;2048,16384,4096;4096,16384,4096;2048,8192,8192;2048,28672,8192;2048,8192,28672;2048,5120,5120;2048,13824,5120;2048,5120,13824"
SHAPES="${S#;}" python scripts/experiments/q256_ab.py time
