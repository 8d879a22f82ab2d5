#!/usr/bin/env python3
"""Average the per-dispatch PMC values of the fused GEMM kernel over the counter passes in a directory written by
scripts/pmc_gpu.sh and print derived ratios."""
import collections
import csv
import glob
import os
import sys

d = sys.argv[1]
vals = collections.defaultdict(list)
for f in glob.glob(os.path.join(d, "p*", "*", "*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "k_qgemm" in r["Kernel_Name"] or "k_mxgemm" in r["Kernel_Name"]:
            vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
avg = {k: sum(v) / len(v) for k, v in vals.items()}
for k in sorted(avg):
    print("%-28s %.4g" % (k, avg[k]))
g = avg.get
if g("SQ_BUSY_CYCLES") and g("SQ_VALU_MFMA_BUSY_CYCLES"):
    print("MFMA busy / SQ busy cycles      %.3f" % (g("SQ_VALU_MFMA_BUSY_CYCLES") / g("SQ_BUSY_CYCLES")))
if g("SQ_WAVE_CYCLES"):
    for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_WAIT_INST_LDS"):
        if g(k):
            print("%-28s / wave cycles %.3f" % (k, g(k) / g("SQ_WAVE_CYCLES")))
if g("SQ_LDS_IDX_ACTIVE") and g("SQ_LDS_BANK_CONFLICT") is not None:
    print("LDS bank conflict / active      %.3f" % (g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE")))
if g("TCC_HIT_sum") and g("TCC_MISS_sum") is not None:
    print("L2 hit rate                     %.3f" % (g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum"))))
if g("SQ_INSTS_VALU") and g("SQ_INSTS_MFMA"):
    # SQ_INSTS_VALU counts the MFMAs too (U8 kernel: 32 converts + 8 shifts per 64 MFMAs in the ISA = 0.6-0.9; the counter
    # ratio reads 1.88): report both
    print("VALU (incl. MFMA) / MFMA instr  %.2f" % (g("SQ_INSTS_VALU") / g("SQ_INSTS_MFMA")))
    print("non-MFMA VALU / MFMA instr      %.2f" % (g("SQ_INSTS_VALU") / g("SQ_INSTS_MFMA") - 1.0))
if g("GRBM_GUI_ACTIVE") and g("SQ_VALU_MFMA_BUSY_CYCLES"):
    # GRBM_GUI_ACTIVE is summed over the 8 XCDs; 1024 SIMDs carry the matrix pipes
    cyc = g("GRBM_GUI_ACTIVE") / 8.0
    print("kernel cycles (per XCD)         %.4g" % cyc)
    print("MFMA pipe busy fraction         %.3f  (SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles))" % (g("SQ_VALU_MFMA_BUSY_CYCLES") / (1024.0 * cyc)))
