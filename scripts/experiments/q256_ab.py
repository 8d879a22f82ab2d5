#!/usr/bin/env python3
"""k_qgemm256 (256-row wave tiles, AGPR accumulators placed by hand) against k_qgemm3 (the compiler-allocated 128-row kernel):
bit-identity of the results on the same planes and interleaved timing in ONE process (MSQ_GEMM_256 is read per call).
Usage: python scripts/experiments/q256_ab.py [check] [time] [M N K ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import msq
from msq import qlinear

dev = torch.device("cuda:0")
args = sys.argv[1:]
do_check = "check" in args or not any(a in args for a in ("check", "time"))
do_time = "time" in args or not any(a in args for a in ("check", "time"))


def weights(N, K, seed=0):
    g = torch.Generator(device=dev).manual_seed(seed)
    W = torch.randn(N, K, generator=g, device=dev) * 0.02
    W[torch.rand(N, K, generator=g, device=dev) < 0.005] *= 16.0
    return W


def run(flag, X, P, dt):
    if flag == "d":                                       # the library's own rule
        os.environ.pop("MSQ_GEMM_256", None)
    else:
        os.environ["MSQ_GEMM_256"] = flag
    return qlinear.qlinear(X, P, None, dt)


if do_check:
    bad = 0
    for (M, N, K) in ((2048, 16384, 4096), (2048 - 37, 2048, 64), (300, 512, 128), (256, 256, 192), (1000, 2304, 320), (4096, 4096, 1088), (513, 11008, 4096)):
        for fo in ("posit8_es1", "fp8_e4m3"):
            P = qlinear.pack_weight(weights(N, K, 1), 8, 8, "fp4_e2m1", fo, 2, 32, layout="unified")
            X = torch.randn(M, K, device=dev, generator=torch.Generator(device=dev).manual_seed(2)).to(torch.bfloat16)
            for dt in (torch.float32, torch.bfloat16, torch.float16):
                a = run("0", X, P, dt)
                b = run("1", X, P, dt)
                c8 = run("2", X, P, dt)
                ok = torch.equal(a, b) and torch.equal(a, c8)
                rep = all(torch.equal(run("1", X, P, dt), b) and torch.equal(run("2", X, P, dt), c8) for _ in range(5))
                ref = X.float() @ qlinear.unpack_weight(P).t()
                err = (b.float() - ref).abs().max().item() / (ref.abs().max().item() + 1e-30)
                print(f"M{M} N{N} K{K} {fo:11s} {str(dt)[6:]:9s}: identical to k_qgemm3 {ok}, repeatable {rep}, rel err vs dense {err:.2e}", flush=True)
                bad += (not ok) or (not rep)
    print("CHECK", "FAILED" if bad else "ok", bad)

if do_time:
    if os.environ.get("SHAPES"):                          # SHAPES="M,N,K;M,N,K;..."
        shapes = [tuple(int(v) for v in t.split(",")) for t in os.environ["SHAPES"].split(";")]
    else:
      shapes = [(2048, 16384, 4096), (2048, 12288, 4096), (2048, 22016, 4096), (2048, 4096, 11008), (2048, 4096, 4096), (4096, 4096, 4096), (1024, 12288, 4096), (4096, 16384, 4096), (2048, 8192, 28672)]
    for (M, N, K) in shapes:
        for fo in ("posit8_es1", "fp8_e4m3"):
            P = qlinear.pack_weight(weights(N, K, 1), 8, 8, "fp4_e2m1", fo, 2, 32, layout="unified")
            X = torch.randn(M, K, device=dev).to(torch.bfloat16)
            for _ in range(100):
                run("0", X, P, torch.bfloat16)
            res = {"0": [], "1": [], "2": [], "d": []}
            for rnd in range(6):
                for flag in ("0", "1", "2", "d"):
                    for _ in range(10):
                        run(flag, X, P, torch.bfloat16)
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(30):
                        run(flag, X, P, torch.bfloat16)
                    e1.record()
                    torch.cuda.synchronize()
                    res[flag].append(e0.elapsed_time(e1) / 30 * 1e3)
            fl = 2.0 * M * N * K
            med = lambda k: sorted(res[k])[len(res[k]) // 2]
            ma, mb, mc, md = med("0"), med("1"), med("2"), med("d")
            best = min(ma, mb, mc)
            print(f"M{M} N{N} K{K} {fo:11s}: k_qgemm3 {ma:7.1f} us ({fl/ma/1e6:6.0f} TF) | k_qgemm256 {mb:7.1f} ({fl/mb/1e6:6.0f} TF, x{ma/mb:.3f}) | MF=8 form {mc:7.1f} ({fl/mc/1e6:6.0f} TF, x{ma/mc:.3f}) | default rule {md:7.1f} (x{best/md:.3f} of the best)", flush=True)
