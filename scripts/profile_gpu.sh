#!/bin/bash
# Runs on the GPU box (via gpurun): kernel trace + PMC passes of the bench command.
# Usage: scripts/profile_gpu.sh <tag> [extra bench args]
set -u
TAG=${1:-r01}; shift || true
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline "$@" > $OUT/bench_trace.json 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" > $OUT/bench_fetch.json 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" > $OUT/bench_write.json 2> $OUT/write.err
find $OUT -name "*.csv" | head -20
