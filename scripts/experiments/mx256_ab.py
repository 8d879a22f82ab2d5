#!/usr/bin/env python3
"""k_mxgemm256 (256-row wave tiles, AGPR accumulators placed by hand) against k_mxgemm: bit-identity on the same operands and
interleaved timing in ONE process (MSQ_MX_256 is read per call).  Usage: python scripts/experiments/mx256_ab.py [check] [time]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import msq
from msq import qlinear, quant

dev = torch.device("cuda:0")
args = sys.argv[1:]
do_check = "check" in args or not args
do_time = "time" in args or not args


def weights(N, K, seed=0):
    g = torch.Generator(device=dev).manual_seed(seed)
    W = torch.randn(N, K, generator=g, device=dev) * 0.02
    W[torch.rand(N, K, generator=g, device=dev) < 0.005] *= 16.0
    return W


def operands(N, K):
    W = weights(N, K, 1)
    return {"fp4": qlinear.mx_pack_weight(W, w_fmt="e2m1"), "e3m2": qlinear.mx_pack_weight(W, w_fmt="e3m2"), "e2m3": qlinear.mx_pack_weight(W, w_fmt="e2m3"),
            "e4m3": qlinear.mx_pack_values(quant.outlier_fakequant(W, 8, 8, "fp4_e2m1", "fp8_e4m3", 2, -1, 32)["out"])}


def run(flag, xp, P, dt):
    if flag == "d":                                       # the library's own rule
        os.environ.pop("MSQ_MX_256", None)
    else:
        os.environ["MSQ_MX_256"] = flag
    return qlinear.qlinear_mx_w4a8(xp, P, None, dt)


if do_check:
    bad = 0
    for (M, N, K) in ((2048, 16384, 4096), (2048 - 37, 2048, 128), (300, 512, 256), (256, 256, 384), (1000, 2304, 640), (4096, 4096, 1152)):
        ops = operands(N, K)
        X = torch.randn(M, K, device=dev, generator=torch.Generator(device=dev).manual_seed(2))
        xp = qlinear.mx_pack_act(X)
        for name, P in ops.items():
            for dt in (torch.float32, torch.bfloat16, torch.float16):
                a = run("0", xp, P, dt)
                b = run("1", xp, P, dt)
                c8 = run("2", xp, P, dt)
                ok = torch.equal(a, b) and torch.equal(b, c8)
                rep = all(torch.equal(run("1", xp, P, dt), b) and torch.equal(run("2", xp, P, dt), c8) for _ in range(5))
                print(f"M{M} N{N} K{K} {name:5s} {str(dt)[6:]:9s}: identical to k_mxgemm {ok}, repeatable {rep}, max|diff| {(a.float() - b.float()).abs().max().item():.3e} of {a.float().abs().max().item():.3e}", flush=True)
                bad += (not ok) or (not rep)
    print("CHECK", "FAILED" if bad else "ok", bad)

if do_time:
    shapes = ((2048, 16384, 4096), (4096, 16384, 4096), (2048, 22016, 4096), (2048, 8192, 28672))
    if os.environ.get("SHAPES"):                          # SHAPES="M,N,K;M,N,K;..."
        shapes = [tuple(int(v) for v in t.split(",")) for t in os.environ["SHAPES"].split(";")]
    for (M, N, K) in shapes:
        ops = operands(N, K)
        if os.environ.get("OPS"):
            ops = {k: v for k, v in ops.items() if k in os.environ["OPS"].split(",")}
        X = torch.randn(M, K, device=dev)
        xp = qlinear.mx_pack_act(X)
        for name, P in ops.items():
            for _ in range(100):
                run("0", xp, P, torch.bfloat16)
            res = {"0": [], "1": [], "2": [], "d": []}
            for rnd in range(6):
                for flag in ("0", "1", "2", "d"):
                    for _ in range(10):
                        run(flag, xp, P, torch.bfloat16)
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(30):
                        run(flag, xp, P, torch.bfloat16)
                    e1.record()
                    torch.cuda.synchronize()
                    res[flag].append(e0.elapsed_time(e1) / 30 * 1e3)
            fl = 2.0 * M * N * K
            ma, mb, mc, md = sorted(res["0"])[3], sorted(res["1"])[3], sorted(res["2"])[3], sorted(res["d"])[3]
            print(f"M{M} N{N} K{K} {name:5s}: k_mxgemm median {ma:7.1f} us ({fl/ma/1e6:6.0f} TF = {fl/ma/1e6/5000:.3f}) | k_mxgemm256 median {mb:7.1f} us ({fl/mb/1e6:6.0f} TF = {fl/mb/1e6/5000:.3f})  ratio {ma/mb:.3f} | MF=8 form {mc:7.1f} us ({fl/mc/1e6/5000:.3f}) ratio {ma/mc:.3f} | default rule {md:7.1f} (x{min(ma, mb, mc)/md:.3f} of the best)", flush=True)
