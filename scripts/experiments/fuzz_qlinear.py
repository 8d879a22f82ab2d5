#!/usr/bin/env python3
"""Randomised differential run of the fused GEMM entry points over the dispatch rules: random (M, N, K), layouts, outlier formats,
output dtypes and bias; the library's own kernel choice must (a) repeat bit for bit, (b) agree with the float64 product of the
unpacked operands within the tolerance of the fixed tests, (c) equal the result with the hand-allocated kernels disabled whenever the
K-summation order is the same (no split-K on either side: checked by equality, reported otherwise).  Seeded; prints failures only + a summary.
Usage: python scripts/experiments/fuzz_qlinear.py [cases] [seed]"""
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import msq
from msq import qlinear, quant

dev = torch.device("cuda:0")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
L = msq._lib.lib()
NAMES = {0: "decode", 1: "gemm128", 2: "t256", 3: "t128", 4: "persistent", 5: "streamk(sk)"}
bad = 0
seen = {}
for case in range(cases):
    if os.environ.get("BIG"):                               # prefill-size grids: the hand-allocated kernels in both block heights
        M = rnd.choice([300, 384, 512, 513, 640, 777, 1000, 1024, 1500, 1990, 2048, 2500, 3072, 4096, 5000])
        N = 256 * rnd.choice([8, 9, 16, 17, 20, 24, 32, 43, 48, 54, 64, 86])
    else:
        M = rnd.choice([1, 7, 16, 17, 33, 64, 65, 100, 128, 200, 256, 300, 384, 512, 640, 777, 1024, 1500, 2048, 2500, 3072, 4096])
        N = 256 * rnd.choice([1, 2, 3, 5, 8, 9, 16, 17, 24, 32, 43, 48, 64, 86])
    K = 64 * rnd.choice([1, 2, 3, 4, 5, 8, 9, 16, 17, 32, 64]) if rnd.random() < 0.7 else 128 * rnd.choice([1, 2, 3, 5, 8, 32])
    if M * N * K > (6e11 if os.environ.get("BIG") else 2.2e11):
        continue
    g = torch.Generator(device=dev).manual_seed(case)
    W = torch.randn(N, K, device=dev, generator=g) * 0.02
    W[torch.rand(N, K, device=dev, generator=g) < 0.01] *= 16
    X = torch.randn(M, K, device=dev, generator=g)
    bias = torch.randn(N, device=dev, generator=g) if rnd.random() < 0.5 else None
    dt = rnd.choice([torch.float32, torch.bfloat16, torch.float16])
    path = rnd.choice(["posit", "fp8", "planes", "mx_fp4", "mx_e4m3", "mx_e3m2"])
    if path.startswith("mx") and K % 128:
        K = (K // 128 + 1) * 128
        W = torch.randn(N, K, device=dev, generator=g) * 0.02
        X = torch.randn(M, K, device=dev, generator=g)
    try:
        if path.startswith("mx"):
            if path == "mx_e4m3":
                Wq = quant.outlier_fakequant(W, 8, 8, "fp4_e2m1", "fp8_e4m3", 2, -1, 32)["out"]
                P = qlinear.mx_pack_values(Wq)
            else:
                ef = {"mx_fp4": ("fp4_e2m1", "e2m1"), "mx_e3m2": ("fp6_e3m2", "e3m2")}[path]
                Wq = msq.mx_ops._quantize_mx(W, 8, ef[0], axes=[-1], block_size=32)
                P = qlinear.mx_pack_weight(W, w_fmt=ef[1])
            xp = qlinear.mx_pack_act(X)
            Xq = msq.mx_ops._quantize_mx(X, 8, "fp8_e4m3", axes=[-1], block_size=32)
            ref = Xq.double() @ Wq.double().t()
            bound = (Xq.abs() @ Wq.abs().t()).double() * 2.0 ** -11 + 1e-6
            if bias is not None:
                ref = ref + bias.double()
            if dt != torch.float32:
                bound = bound + 2.0 ** -8 * ref.abs()            # the output rounding itself (bf16: half an ulp = 2^-9 relative)
            f = lambda: qlinear.qlinear_mx_w4a8(xp, P, bias, dt)
            choice = L.msq_qlinear_kernel_choice(M, N, K, 0, {"mx_fp4": 0, "mx_e4m3": 1, "mx_e3m2": 3}[path])
            env = "MSQ_MX_256"
        else:
            fo = {"posit": "posit8_es1", "fp8": "fp8_e4m3", "planes": "fp8_e4m3"}[path]
            P = qlinear.pack_weight(W, 8, 8, "fp4_e2m1", fo, 2, 32, layout="planes" if path == "planes" else "unified")
            Xb = X.to(torch.bfloat16)
            ref = Xb.double() @ qlinear.unpack_weight(P).double().t()
            bound = torch.full_like(ref, 2e-5 * ref.abs().max().item() + 1e-6)
            if bias is not None:
                ref = ref + bias.double()
            if dt != torch.float32:
                bound = bound + 2.0 ** -8 * ref.abs()
            f = lambda: qlinear.qlinear(Xb, P, bias, dt)
            choice = L.msq_qlinear_kernel_choice(M, N, K, P.out_kind, -1)
            env = "MSQ_GEMM_256"
        os.environ.pop(env, None)
        y = f()
        rep = torch.equal(f(), y) and torch.equal(f(), y)
        ok = bool(((y.double() - ref).abs() <= bound).all().item())
        os.environ[env] = "0"
        y0 = f()
        os.environ.pop(env, None)
        same = torch.equal(y0, y)
        seen[(path.split("_")[0], NAMES.get(choice, choice))] = seen.get((path.split("_")[0], NAMES.get(choice, choice)), 0) + 1
        if not (ok and rep):
            bad += 1
            print(f"FAIL case {case}: {path} M{M} N{N} K{K} {str(dt)[6:]} bias {bias is not None} kernel {NAMES.get(choice, choice)}: within bound {ok}, repeatable {rep}, max err {(y.double() - ref).abs().max().item():.3e}", flush=True)
        elif not same and choice in (2, 3):
            ks_note = "(other K-summation order on the 128-row side: split-K / 64-row blocks)"
            if not bool(((y0.double() - ref).abs() <= bound).all().item()):
                bad += 1
                print(f"FAIL case {case}: the 128-row result itself is out of bound: {path} M{M} N{N} K{K}", flush=True)
    except Exception as e:                                   # noqa
        bad += 1
        print(f"ERROR case {case}: {path} M{M} N{N} K{K} {str(dt)[6:]}: {type(e).__name__}: {str(e)[:160]}", flush=True)
print("kernel families exercised:", dict(sorted(seen.items())))
print("FUZZ", "FAILED" if bad else "ok", bad, "of", cases)
sys.exit(1 if bad else 0)
