#!/usr/bin/env python3
"""Which bound of the packed in-dtype fake-quant turns blocks away (debug build: scripts/experiments/build_q256.sh -s msq_quant_lowp why
"-ffp-contract=off -DMSQ_LOWP_WHY"; MSQ_LIB_OVERRIDE=scripts/experiments/abl/libmsq_hip_quant_lowp_why.so).  Counts are per BLOCK and per reason
(a block can fail several); slot 15 = blocks seen, 14 = blocks turned away."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import lowp_pk_fuzz as F
import torch
NAMES = ["lo/hi NaN", "non-finite member", "e_in NaN (scale range)", "e_in < lo", "e_in > hi", "e_out NaN", "e_out < lo", "e_out > hi", "fp16 tie_in < -22", "fp16 in top > 15",
         "fp16 e_in > 0 & raw tie", "fp16 out top > 15", "fp16 e_out > 0 & raw tie", "fp16 bound not a half", "turned away", "blocks"]
L = F.L
buf = (ctypes.c_ulonglong * 16)()
g = torch.Generator(device=F.dev).manual_seed(7)
for dt in (torch.float16, torch.bfloat16):
    for kind in sys.argv[1:] or ["weights", "ties", "scales", "sparse", "negative", "subnormal"]:
        W = F.make(kind, tuple(int(v) for v in os.environ.get("WHY_SHAPE", "2048,1024").split(",")), dt, g)
        for fi, fo in (("int2", "fp4"), ("fp4_e2m1", "fp8_e4m3")):
            for axis, bs in ((0, 16), (-1, 32)):
                L.msq_lowp_why_(buf, 1)
                F.run(W, fi, fo, 2.0, axis, bs, 8, 1)
                torch.cuda.synchronize()
                L.msq_lowp_why_(buf, 1)
                v = list(buf)
                print(str(dt)[6:], kind, fi, fo, "axis", axis, "bs", bs, ": blocks", v[15], "turned away", v[14], "|", ", ".join("%s %d" % (NAMES[i], v[i]) for i in range(14) if v[i]))
