#!/bin/bash
# Runs on the GPU box (through gpurun): the whole evidence set of a round.  Usage: scripts/round_final.sh <rNN>
set -u
R=${1:-r04}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/final_$R; mkdir -p $O
python3 -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|error" | tail -3 > $O/gpu_tests.txt
python3 bench.py > $O/${R}_bench_posit.json 2> $O/bench.err
python3 bench.py --outlier fp8_e4m3 --no-cpu-baseline > $O/${R}_bench_fp8.json 2>> $O/bench.err
scripts/profile_gpu.sh ${R}_posit > /dev/null 2>&1
scripts/profile_gpu.sh ${R}_fp8 --outlier fp8_e4m3 > /dev/null 2>&1
scripts/pmc_gpu.sh ${R}_posit > $O/${R}_pmc_posit.txt 2>&1
scripts/pmc_gpu.sh ${R}_fp8 --outlier fp8_e4m3 > $O/${R}_pmc_fp8.txt 2>&1
scripts/round_profiles.sh $R > /dev/null 2>&1
scripts/experiments/pmc_one.sh act1 k_act_quant_rows > $O/${R}_pmc_act_quant_v1.txt 2>&1
python3 scripts/experiments/producers_time.py 2> /dev/null > $O/${R}_producers_run.txt
python3 scripts/experiments/kv_mx_time.py 2> /dev/null > $O/${R}_kv_mx_run.txt
cat $O/gpu_tests.txt
