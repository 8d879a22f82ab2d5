#!/usr/bin/env python3
"""k_qgemm256p (persistent blocks, stream-K over the part-filled last round; msq_gemm256p.hip) against the non-persistent kernels:
results (bit-identical where no tile is cut, within fp32 rounding of the uncut sum where tiles are cut, identical from run to run
either way) and interleaved timing in ONE process (MSQ_GEMM_256 is read per call: 3 forces the persistent kernel, unset = the rules).
Usage: python scripts/experiments/qp_ab.py [check] [time] ;  SHAPES="M,N,K;M,N,K" restricts the timing table."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import msq
from msq import qlinear
from msq._lib import lib

dev = torch.device("cuda:0")
args = sys.argv[1:]
do_check = "check" in args or not any(a in args for a in ("check", "time"))
do_time = "time" in args or not any(a in args for a in ("check", "time"))


def weights(N, K, seed=0):
    g = torch.Generator(device=dev).manual_seed(seed)
    W = torch.randn(N, K, generator=g, device=dev) * 0.02
    W[torch.rand(N, K, generator=g, device=dev) < 0.005] *= 16.0
    return W


def run(flag, X, P, dt, bias=None):
    if flag == "d":
        os.environ.pop("MSQ_GEMM_256", None)
    else:
        os.environ["MSQ_GEMM_256"] = flag
    return qlinear.qlinear(X, P, bias, dt)


def plan(M, N, K):
    v = [ctypes.c_int(0) for _ in range(4)]
    ws = ctypes.c_int64(0)
    rc = lib().msq_qgemm256p_plan(ctypes.c_int64(M), ctypes.c_int64(N), ctypes.c_int64(K), 0, *[ctypes.byref(x) for x in v], ctypes.byref(ws))
    return rc, [x.value for x in v], ws.value


if do_check:
    bad = 0
    cases = ((2048, 16384, 4096), (2048, 12288, 4096), (2048, 4096, 4096), (2048, 22016, 4096), (2048, 4096, 11008), (2048 - 37, 2048, 128),
             (300, 512, 256), (256, 256, 384), (1000, 2304, 640), (4096, 4096, 1152), (513, 11008, 4096), (777, 16384, 4096), (4096, 11008, 4096))
    for (M, N, K) in cases:
        rc, (Pb, full, R, q), wsb = plan(M, N, K)
        for fo in ("posit8_es1", "fp8_e4m3"):
            P = qlinear.pack_weight(weights(N, K, 1), 8, 8, "fp4_e2m1", fo, 2, 32, layout="unified")
            X = torch.randn(M, K, device=dev, generator=torch.Generator(device=dev).manual_seed(2)).to(torch.bfloat16)
            bias = torch.randn(N, device=dev)
            for dt in (torch.float32, torch.bfloat16):
                for bb in (None, bias):
                    a = run("1", X, P, dt, bb)                       # k_qgemm256, 256-row blocks
                    b = run("3", X, P, dt, bb)
                    cut = wsb > 0
                    same = torch.equal(a, b)
                    ref = X.float() @ qlinear.unpack_weight(P).t() + (0 if bb is None else bb)
                    sc = ref.abs().max().item() + 1e-30
                    err = (b.float() - ref).abs().max().item() / sc
                    erra = (a.float() - ref).abs().max().item() / sc
                    dab = (a.float() - b.float()).abs().max().item() / sc
                    rep = all(torch.equal(run("3", X, P, dt, bb), b) for _ in range(6))
                    ok = rep and (same if not cut else (dab <= (2e-5 if dt == torch.float32 else 8e-3) and err <= max(2 * erra, 2e-5)))
                    print(f"M{M} N{N} K{K} plan P{Pb} full{full} R{R} q{q} ws{wsb >> 20}MB {fo:11s} {str(dt)[6:]:9s} bias {bb is not None!s:5s}: "
                          f"{'identical' if same else 'max diff %.2e' % dab}, repeatable {rep}, rel err vs dense {err:.2e} (k_qgemm256 {erra:.2e}) {'ok' if ok else 'FAIL'}", flush=True)
                    bad += not ok
    print("CHECK", "FAILED" if bad else "ok", bad)

if do_time:
    if os.environ.get("SHAPES"):
        shapes = [tuple(int(v) for v in t.split(",")) for t in os.environ["SHAPES"].split(";")]
    else:
        shapes = [(2048, 16384, 4096), (2048, 12288, 4096), (2048, 4096, 4096), (2048, 22016, 4096), (2048, 4096, 11008), (4096, 4096, 4096),
                  (4096, 16384, 4096), (512, 22016, 4096), (2048, 8192, 28672)]
    for (M, N, K) in shapes:
        for fo in ("posit8_es1", "fp8_e4m3"):
            P = qlinear.pack_weight(weights(N, K, 1), 8, 8, "fp4_e2m1", fo, 2, 32, layout="unified")
            X = torch.randn(M, K, device=dev).to(torch.bfloat16)
            Wd = qlinear.unpack_weight(P, torch.bfloat16)
            for _ in range(100):
                run("d", X, P, torch.bfloat16)
            res = {"d": [], "3": [], "h": []}
            for rnd in range(6):
                for flag in ("d", "3", "h"):
                    fn = (lambda: X @ Wd.t()) if flag == "h" else (lambda: run(flag, X, P, torch.bfloat16))
                    for _ in range(10):
                        fn()
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(30):
                        fn()
                    e1.record()
                    torch.cuda.synchronize()
                    res[flag].append(e0.elapsed_time(e1) / 30 * 1e3)
            fl = 2.0 * M * N * K
            med = lambda k: sorted(res[k])[len(res[k]) // 2]
            md, mp, mh = med("d"), med("3"), med("h")
            print(f"M{M} N{N} K{K} {fo:11s}: rules {md:7.1f} us ({fl / md / 1e6 / 2500:.3f})  persistent {mp:7.1f} us ({fl / mp / 1e6 / 2500:.3f})  "
                  f"hipBLASLt bf16 {mh:7.1f} us ({fl / mh / 1e6 / 2500:.3f})   persistent / rules {md / mp:.3f}", flush=True)
    os.environ.pop("MSQ_GEMM_256", None)
