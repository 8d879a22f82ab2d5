#!/usr/bin/env python3
"""Narrow projections (and the two-plane layout) at decode sizes: split-K planes with the fix-up fused into the same launch
(MSQ_GEMV_FIXUP=1: the last block of a strip sums the planes) against planes + the k_splitk_reduce launch (MSQ_GEMV_FIXUP=0).
(EXPERIMENT: needs scripts/experiments/decode_fused_fixup.patch applied to csrc/msq_gemm.hip -- measured slower, not shipped; see
profiles/r04_decode_narrow_experiments.txt.)  `check`: equal bit for bit, eagerly and from captured graphs; `time`: COLD weights (>= 1.2 GB of distinct copies per graph, as
decode_cold.py).  Usage: python scripts/experiments/decode_narrow_ab.py [check] [time]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import msq
from msq import qlinear
from decode_cold import clone, tgraph  # noqa: E402  (same directory)

dev = torch.device("cuda:0")
args = sys.argv[1:]
FO = {"fp8": "fp8_e4m3", "posit": "posit8_es1"}
SH = (("o", 4096, 4096), ("down", 4096, 11008))


def run(flag, X, P, bias, dt):
    os.environ["MSQ_GEMV_FIXUP"] = flag
    return qlinear.qlinear(X, P, bias, dt)


if "check" in args:
    bad = 0
    for (name, N, K) in SH + (("odd", 2048, 4096 + 64), ("n2304", 2304, 1024), ("wide-planes", 12288, 2048)):
        W = torch.randn(N, K, device=dev) * 0.02
        W[torch.rand(N, K, device=dev) < 0.005] *= 16
        for f, layout in (("fp8", "unified"), ("posit", "unified"), ("fp8", "planes")):
            P = qlinear.pack_weight(W, 8, 8, "fp4_e2m1", FO[f], 2, 32, layout=layout)
            bias = torch.randn(N, device=dev)
            for M in (1, 3, 16, 17, 32, 48):
                Xs = [torch.randn(M, K, device=dev).to(torch.bfloat16) for _ in range(3)]      # the input changes from launch to launch: a stale
                for dt in (torch.float32, torch.bfloat16, torch.float16):                      # partial tile of the launch before would show
                    ref = [run("0", X, P, bias, dt) for X in Xs]
                    ok = all(torch.equal(run("1", Xs[i % 3], P, bias, dt), ref[i % 3]) for i in range(60))
                    # from a captured graph, replayed: the counters return to zero every time
                    os.environ["MSQ_GEMV_FIXUP"] = "1"
                    s = torch.cuda.Stream()
                    xin = Xs[0].clone()
                    with torch.cuda.stream(s):
                        g = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(g, stream=s):
                            y = qlinear.qlinear(xin, P, bias, dt)
                    for i in range(30):
                        xin.copy_(Xs[i % 3])
                        g.replay()
                        torch.cuda.synchronize()
                        ok = ok and torch.equal(y, ref[i % 3])
                    bad += not ok
                    if not ok or (M in (1, 32) and dt == torch.float32):
                        print(f"{name} N{N} K{K} {f} {layout} M{M} {str(dt)[6:]}: fused fix-up == planes + reduce launch: {ok}", flush=True)
    print("CHECK", "FAILED" if bad else "ok", bad)

if "time" in args:
    for (name, N, K) in SH:
        W = torch.randn(N, K, device=dev) * 0.02
        W[torch.rand(N, K, device=dev) < 0.005] *= 16
        for f in ("fp8", "posit"):
            P0 = qlinear.pack_weight(W, 8, 8, "fp4_e2m1", FO[f], 2, 32, layout="unified")
            copies = max(4, int(1.2e9 // P0.nbytes) + 1)
            Ps = [P0] + [clone(P0) for _ in range(copies - 1)]
            for M in (1, 4, 8, 16, 32):
                X = torch.randn(M, K, device=dev).to(torch.bfloat16)
                out = []
                for flag in ("0", "1"):
                    os.environ["MSQ_GEMV_FIXUP"] = flag
                    us = min(tgraph([(lambda P=P: qlinear.qlinear(X, P)) for P in Ps]) for _ in range(3))
                    out.append(("planes + reduce launch" if flag == "0" else "fused fix-up") + f" {us:5.1f} us")
                print(f"{name:5s} {f:5s} M{M:3d} ({P0.nbytes/1e6:.0f} MB): " + " | ".join(out), flush=True)
            del Ps
