#!/usr/bin/env python3
"""128 x 128 blocks of two waves (k_qgemm256<., ., 8, 2>, optional split-K with fp32 planes) against the library's choice for grids between
decode and prefill sizes: check against the dense product and run-to-run identity, interleaved timing in ONE process
(MSQ_GEMM_N128_KS is read per call).  Usage: python scripts/experiments/n128_ab.py [check] [time]   (SHAPES="M,N,K;..." overrides)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import msq
from msq import qlinear

dev = torch.device("cuda:0")
args = sys.argv[1:]


def weights(N, K, seed=0):
    g = torch.Generator(device=dev).manual_seed(seed)
    W = torch.randn(N, K, generator=g, device=dev) * 0.02
    W[torch.rand(N, K, generator=g, device=dev) < 0.005] *= 16.0
    return W


def run(ks, X, P, bias, dt):
    if ks is None:
        os.environ.pop("MSQ_GEMM_N128_KS", None)
        os.environ["MSQ_GEMM_N128_OFF"] = "1"
    else:
        os.environ.pop("MSQ_GEMM_N128_OFF", None)
        os.environ["MSQ_GEMM_N128_KS"] = str(ks)
    return qlinear.qlinear(X, P, bias, dt)


if "check" in args:
    bad = 0
    for (M, N, K) in ((128, 16384, 4096), (100, 2304, 512), (129, 512, 1024), (300, 4096, 4160), (65, 256, 256), (256, 12288, 4096)):
        for fo in ("posit8_es1", "fp8_e4m3"):
            P = qlinear.pack_weight(weights(N, K, 1), 8, 8, "fp4_e2m1", fo, 2, 32, layout="unified")
            X = torch.randn(M, K, device=dev, generator=torch.Generator(device=dev).manual_seed(2)).to(torch.bfloat16)
            bias = torch.randn(N, device=dev)
            ref = X.double() @ qlinear.unpack_weight(P).double().t() + bias.double()
            for dt in (torch.float32, torch.bfloat16, torch.float16):
                a = run(None, X, P, bias, dt)
                for ks in (1, 2, 4):
                    if ks * 4 > K // 64:
                        continue
                    b = run(ks, X, P, bias, dt)
                    rep = all(torch.equal(run(ks, X, P, bias, dt), b) for _ in range(5))
                    tol = (2e-5 if dt == torch.float32 else 8e-3) * ref.abs().max().item()
                    e = (b.double() - ref).abs().max().item()
                    same = torch.equal(a, b)
                    ok = e <= tol and rep
                    bad += not ok
                    if not ok or dt == torch.float32:
                        print(f"M{M} N{N} K{K} {fo:11s} {str(dt)[6:]:9s} ks{ks}: err {e:.2e} (tol {tol:.2e}), repeatable {rep}, equal to the library's choice {same} {'ok' if ok else 'BAD'}", flush=True)
    print("CHECK", "FAILED" if bad else "ok", bad)

if "time" in args:
    shapes = [(m, n, k) for m in (65, 96, 128, 192, 256, 384) for (n, k) in ((12288, 4096), (4096, 4096), (22016, 4096), (4096, 11008), (16384, 4096))]
    if os.environ.get("SHAPES"):
        shapes = [tuple(int(v) for v in t.split(",")) for t in os.environ["SHAPES"].split(";")]
    for (M, N, K) in shapes:
        for fo in ("posit8_es1", "fp8_e4m3"):
            P = qlinear.pack_weight(weights(N, K, 1), 8, 8, "fp4_e2m1", fo, 2, 32, layout="unified")
            X = torch.randn(M, K, device=dev).to(torch.bfloat16)
            Wd = qlinear.unpack_weight(P, torch.bfloat16)
            confs = [None, 1, 2, 4]
            res = {c: [] for c in confs + ["dense"]}
            for _ in range(50):
                run(None, X, P, None, torch.bfloat16)
            for rnd in range(5):
                for c in confs + ["dense"]:
                    f = (lambda: X @ Wd.t()) if c == "dense" else (lambda: run(c, X, P, None, torch.bfloat16))
                    for _ in range(10):
                        f()
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(30):
                        f()
                    e1.record()
                    torch.cuda.synchronize()
                    res[c].append(e0.elapsed_time(e1) / 30 * 1e3)
            med = {c: sorted(v)[len(v) // 2] for c, v in res.items()}
            best = min((1, 2, 4), key=lambda c: med[c])
            print(f"M{M} N{N} K{K} {fo:11s}: library {med[None]:6.1f} us | 128x128 blocks ks1 {med[1]:6.1f} ks2 {med[2]:6.1f} ks4 {med[4]:6.1f} (best ks{best}: x{med[None]/med[best]:.3f}) | hipBLASLt bf16 dense {med['dense']:6.1f}", flush=True)
