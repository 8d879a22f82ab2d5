#!/usr/bin/env python3
"""Decode (M = 1 ... 16) time of the fused Llama-2-7B projections with COLD weights: every graph walks through enough distinct
copies of the packed weight (>= 1 GB) that neither the L2 nor the 256 MB Infinity Cache can hold them -- what a decode step
over 32 layers sees (a graph replaying ONE weight measures the Infinity Cache instead).
Usage: python scripts/experiments/decode_cold.py [fp8|posit] [M ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import msq
from msq import qlinear

dev = torch.device("cuda:0")
torch.manual_seed(0)
fmts = [a for a in sys.argv[1:] if a in ("fp8", "posit", "mx8", "mx4")] or ["fp8", "posit"]      # mx8 / mx4: the MX matrix path (e4m3 values / plain MX-FP4 operand)
Ms = [int(a) for a in sys.argv[1:] if a.isdigit()] or [1, 16]
FO = {"fp8": "fp8_e4m3", "posit": "posit8_es1"}


def clone(P):
    c = lambda t: None if t is None else t.clone()
    if isinstance(P, qlinear.MXPackedWeight):
        import copy
        Q = copy.copy(P)
        Q.codes, Q.scales = P.codes.clone(), P.scales.clone()
        return Q
    return qlinear.PackedWeight(c(P.inl), c(P.out), c(P.scl), P.N, P.K, P.block, P.in_kind, P.out_kind, P.n, P.k)


def tgraph(fns, reps=5):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for f in fns[:2]:
            f()
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for f in fns:
                f()
    for _ in range(2):
        g.replay()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps / len(fns) * 1e3


def main():
    tot = {}
    ONLY = [a for a in os.environ.get("ONLY", "").split(",") if a]
    for (name, N, K) in (("qkv", 12288, 4096), ("o", 4096, 4096), ("gate_up", 22016, 4096), ("down", 4096, 11008)):
        if ONLY and name not in ONLY:
            continue
        W = torch.randn(N, K, device=dev) * 0.02
        W[torch.rand(N, K, device=dev) < 0.005] *= 16
        for f in fmts:
            if f == "mx8":
                P0 = qlinear.mx_pack_values(msq.quant.outlier_fakequant(W[:, :K // 128 * 128].contiguous(), 8, 8, "fp4_e2m1", "fp8_e4m3", 2, -1, 32)["out"])
            elif f == "mx4":
                P0 = qlinear.mx_pack_weight(W[:, :K // 128 * 128].contiguous())
            else:
                P0 = qlinear.pack_weight(W, 8, 8, "fp4_e2m1", FO[f], 2, 32, layout="unified")
            call = qlinear.qlinear_mx_w4a8 if f in ("mx8", "mx4") else qlinear.qlinear
            copies = max(4, int(1.2e9 // P0.nbytes) + 1)
            Ps = [P0] + [clone(P0) for _ in range(copies - 1)]
            for M in Ms:
                X = torch.randn(M, P0.k, device=dev).to(torch.bfloat16)
                us = tgraph([(lambda P=P: call(X, P)) for P in Ps])
                tot[(f, M)] = tot.get((f, M), 0.0) + us
                print(f"{name:8s} N{N:6d} K{K:6d} {f:5s} M{M:3d}: {us:6.1f} us  {P0.nbytes/us/1e3:5.0f} GB/s of packed weight ({copies} copies, {P0.nbytes/1e6:.0f} MB each)", flush=True)
            del Ps
        if os.environ.get("DENSE", "0") == "1":
            Wb = W.to(torch.bfloat16)
            copies = max(4, int(1.2e9 // (N * K * 2)) + 1)
            Ws = [Wb] + [Wb.clone() for _ in range(copies - 1)]
            for M in Ms:
                X = torch.randn(M, K, device=dev).to(torch.bfloat16)
                us = tgraph([(lambda w=w: X @ w.t()) for w in Ws])
                print(f"{name:8s} N{N:6d} K{K:6d} dense bf16 hipBLASLt M{M:3d}: {us:6.1f} us  {N*K*2/us/1e3:5.0f} GB/s", flush=True)
            del Ws
        del W
    for k, v in sorted(tot.items()):
        print(f"layer total {k[0]} M{k[1]}: {v:6.1f} us")


if __name__ == "__main__":
    main()
