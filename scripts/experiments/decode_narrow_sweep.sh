#!/bin/bash
# cold-weight decode of the narrow projections (o 4096 x 4096, down 4096 x 11008) against the wave-count target of the split-K
# decode kernel (MSQ_GEMV_TARGET_WAVES, read once per process).  Usage (GPU box): scripts/experiments/decode_narrow_sweep.sh
cd "$GRAFT_REPO_ROOT"
for tw in 768 1024 1536 2048 3072 4096 6144; do
  echo "== MSQ_GEMV_TARGET_WAVES=$tw"
  MSQ_GEMV_TARGET_WAVES=$tw ONLY=o,down python3 scripts/experiments/decode_cold.py posit 1 2>&1 | grep -v amdgpu.ids | grep "^o \|^down"
done
