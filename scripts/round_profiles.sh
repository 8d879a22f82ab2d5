#!/bin/bash
# Runs on the GPU box (through gpurun): the measurements the per-round profile set is built from.
# Usage: scripts/round_profiles.sh <rNN> [part ...]   parts: timings decode side mx w4a8 rowpar prefill e2e   (default: all)
set -u
R=${1:-r04}; shift || true
PARTS=${*:-timings decode side mx w4a8 rowpar prefill e2e e2emx}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/round_$R; mkdir -p $O
for p in $PARTS; do
  case $p in
    timings) python3 scripts/measure_kernels.py > $O/${R}_kernel_timings.txt 2> $O/timings.err ;;
    decode)  DENSE=1 python3 scripts/experiments/decode_cold.py 1 16 32 2> $O/decode.err | grep -v amdgpu.ids > $O/${R}_decode_cold.txt ;;
    side)    python3 scripts/experiments/side_time.py 2> $O/side.err | grep -v amdgpu.ids > $O/${R}_side_kernels.txt ;;
    mx)      for w in mx_w4a8 msq_w4a8_mx mx_w6a8; do scripts/profile_gpu.sh ${R}_$w --workload llama7b_$w > /dev/null 2>&1; python3 bench.py --workload llama7b_$w --no-cpu-baseline 2> /dev/null > $O/${R}_bench_$w.json; done ;;
    w4a8)    python3 bench.py --workload llama7b_w4a8 --no-cpu-baseline 2> /dev/null > $O/${R}_bench_w4a8.json ;;
    rowpar)  python3 bench.py --workload llama70b_rowparallel --no-cpu-baseline 2> /dev/null > $O/${R}_bench_70b_rowparallel_1gpu.json
             python3 bench.py --workload llama70b_rowparallel --mx --no-cpu-baseline 2> /dev/null > $O/${R}_bench_70b_rowparallel_mx_1gpu.json ;;
    prefill) python3 scripts/experiments/layer_prefill.py 2> $O/prefill.err | grep -v amdgpu.ids > $O/${R}_layer_prefill.txt ;;
    e2e)     python3 bench.py --workload llama7b_e2e --no-cpu-baseline 2> /dev/null > $O/${R}_bench_e2e.json
             python3 bench.py --workload llama7b_e2e --model-dtype bf16 --no-cpu-baseline 2> /dev/null > $O/${R}_bench_e2e_bf16model.json ;;
    e2emx)   python3 bench.py --workload llama7b_e2e --model-dtype bf16 --outlier fp8_e4m3 --path mx --no-cpu-baseline 2> /dev/null > $O/${R}_bench_e2e_bf16model_mx_fp8.json
             python3 bench.py --workload llama7b_e2e --model-dtype fp16 --outlier fp8_e4m3 --path mx --no-cpu-baseline 2> /dev/null > $O/${R}_bench_e2e_fp16model_mx_fp8.json ;;
  esac
done
ls -la $O | tail -20
