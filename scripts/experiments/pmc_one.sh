#!/bin/bash
# PMC passes for one side kernel.  Usage (on the GPU box): scripts/experiments/pmc_one.sh <what> <kernel name substring>
set -u
W=$1; K=$2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc1_$W; mkdir -p $OUT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES SQ_INSTS_VMEM_WR" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" "FETCH_SIZE" ; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -- python3 scripts/experiments/one_kernel.py $W > /dev/null 2> $OUT/p$i.err
done
python3 - "$OUT" "$K" <<'PY'
import csv, glob, sys, collections
out, key = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in glob.glob(out + "/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if key in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
n = None
for k, v in sorted(acc.items()):
    # one row per (dispatch, counter [, dimension]): sum over a dispatch's rows, average over dispatches
    print("%-26s total/launch %.4g" % (k, sum(v) / 10))
PY
