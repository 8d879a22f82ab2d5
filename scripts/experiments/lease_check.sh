#!/bin/bash
# One lease of the pool: the two headline kernels at X[2048,4096] x W[16384,4096]^T (posit / fp8 outliers on the bf16 MFMA; MX-FP4, e4m3 and
# MX-FP6 operands on the scaled MFMA), 256-row forms against the 128-row kernels of round 3, with the clocks rocm-smi reports
# right after.  Usage (on the GPU box): scripts/experiments/lease_check.sh >> profiles/rNN_leases.txt
cd "$(dirname "$0")/../.."
echo "## lease $(date -u +%H:%M:%S) $(rocm-smi --showuniqueid 2>/dev/null | grep "GPU\[" | head -1 | sed 's/.*: *//') $(hostname)"
( for i in 1 2 3 4 5 6; do sleep 6; echo "   (under load) $(rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|Power (W)" | sed 's/GPU\[0\]\t*: *//' | tr '\n' ' ')"; done ) &
SAMPLER=$!
SHAPES="2048,16384,4096" python scripts/experiments/q256_ab.py time 2>&1 | grep "^M" | sed 's/ | MF=8 form.*//'
SHAPES="2048,16384,4096" python scripts/experiments/mx256_ab.py time 2>&1 | grep "^M" | sed 's/ | MF=8 form.*//'
wait $SAMPLER 2>/dev/null
