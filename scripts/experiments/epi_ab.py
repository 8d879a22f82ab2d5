#!/usr/bin/env python3
"""Register-exchange epilogue (v_permlane16_swap, product build) against the LDS-transposed one (abl/libmsq_hip_epilds.so, build_epi.sh)
in k_qgemm256 / k_mxgemm256: same bits on every output element, interleaved timing in ONE process."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import msq
from msq import qlinear, quant, _lib
dev = torch.device("cuda:0"); torch.manual_seed(0)
here = os.path.dirname(os.path.abspath(__file__))
libs = [("direct", _lib.so_path()), ("lds", os.path.join(here, "abl", "libmsq_hip_epilds.so"))]
H = []
for n, p in libs:
    L = C.CDLL(p)
    for f in ("msq_qlinear_bf16", "msq_qlinear_mx_w4a8", "msq_qlinear_mx_w8a8"):
        fn = getattr(L, f); fn.restype = C.c_int; fn.argtypes = _lib._SIGS[f][1]
    H.append((n, L))
YD = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}
def weights(N, K):
    W = torch.randn(N, K, device=dev) * 0.02; W[torch.rand(N, K, device=dev) < 0.005] *= 16; return W
def call_q(L, X, P, b, Y):
    rc = L.msq_qlinear_bf16(_lib.ptr(X), _lib.ptr(P.inl), _lib.ptr(P.out), _lib.ptr(P.scl), _lib.ptr(b), _lib.ptr(Y), YD[Y.dtype], X.shape[0], P.N, P.K, P.block, P.in_kind, P.out_kind, None, 0, _lib.current_stream())
    assert rc == 0, rc
def call_mx(L, fn, Xp, Pw, b, Y):
    rc = getattr(L, fn)(_lib.ptr(Xp[0]), _lib.ptr(Xp[1]), _lib.ptr(Pw.codes), _lib.ptr(Pw.scales), _lib.ptr(b), _lib.ptr(Y), YD[Y.dtype], Xp[0].shape[0], Pw.N, Pw.K, None, 0, _lib.current_stream())
    assert rc == 0, rc
os.environ["MSQ_GEMM_256"] = "1"; os.environ["MSQ_MX_256"] = "1"
bad = 0
if "time" not in sys.argv[1:]:
    for (M, N, K) in ((2048, 4096, 512), (777, 2048, 256), (300, 512, 384), (256, 256, 128)):
        W = weights(N, K); X = torch.randn(M, K, device=dev).to(torch.bfloat16); bias = torch.randn(N, device=dev)
        for fo in ("posit8_es1", "fp8_e4m3"):
            P = qlinear.pack_weight(W, 8, 8, "fp4_e2m1", fo, 2, 32, layout="unified")
            for dt in (torch.float32, torch.float16, torch.bfloat16):
                for b in (None, bias):
                    for mode in ("1", "2"):
                        os.environ["MSQ_GEMM_256"] = mode
                        ys = []
                        for n, L in H:
                            Y = torch.full((M, N), float("nan"), dtype=dt, device=dev); call_q(L, X, P, b, Y); ys.append(Y)
                        ok = torch.equal(ys[0], ys[1]) and not torch.isnan(ys[0]).any().item()
                        bad += not ok
                        print(f"qgemm256 mf{'16' if mode == '1' else ' 8'} M{M} N{N} K{K} {fo:11s} {str(dt)[6:]:9s} bias {b is not None!s:5s}: {'identical' if ok else 'DIFFERENT'}", flush=True)
        if K % 128 == 0:
            Xp = qlinear.mx_pack_act(X.float())
            for tag, Pw, fn in (("mx fp4", qlinear.mx_pack_weight(W), "msq_qlinear_mx_w4a8"),
                                ("mx e4m3", qlinear.mx_pack_values(quant.outlier_fakequant(W, 8, 8, "fp4_e2m1", "fp8_e4m3", 2, -1, 32)["out"]), "msq_qlinear_mx_w8a8")):
                for dt in (torch.float32, torch.bfloat16):
                    for b in (None, bias):
                        for mode in ("1", "2"):
                            os.environ["MSQ_MX_256"] = mode
                            ys = []
                            for n, L in H:
                                Y = torch.full((M, N), float("nan"), dtype=dt, device=dev); call_mx(L, fn, Xp, Pw, b, Y); ys.append(Y)
                            ok = torch.equal(ys[0], ys[1]) and not torch.isnan(ys[0]).any().item()
                            bad += not ok
                            print(f"mxgemm256 mode {mode} M{M} N{N} K{K} {tag:8s} {str(dt)[6:]:9s} bias {b is not None!s:5s}: {'identical' if ok else 'DIFFERENT'}", flush=True)
    print("CHECK", "FAILED" if bad else "ok", bad)
os.environ["MSQ_GEMM_256"] = "1"; os.environ["MSQ_MX_256"] = "1"
for (M, N, K) in ((2048, 16384, 4096), (4096, 4096, 4096)):
    W = weights(N, K); X = torch.randn(M, K, device=dev).to(torch.bfloat16)
    Xp = qlinear.mx_pack_act(X.float())
    cases = [("posit", lambda L, Y, P=qlinear.pack_weight(W, 8, 8, "fp4_e2m1", "posit8_es1", 2, 32, layout="unified"): call_q(L, X, P, None, Y)),
             ("fp8", lambda L, Y, P=qlinear.pack_weight(W, 8, 8, "fp4_e2m1", "fp8_e4m3", 2, 32, layout="unified"): call_q(L, X, P, None, Y)),
             ("mx fp4", lambda L, Y, Pw=qlinear.mx_pack_weight(W): call_mx(L, "msq_qlinear_mx_w4a8", Xp, Pw, None, Y)),
             ("mx e4m3", lambda L, Y, Pw=qlinear.mx_pack_values(quant.outlier_fakequant(W, 8, 8, "fp4_e2m1", "fp8_e4m3", 2, -1, 32)["out"]): call_mx(L, "msq_qlinear_mx_w8a8", Xp, Pw, None, Y))]
    Y = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    for tag, fn in cases:
        res = {n: [] for n, _ in H}
        for _ in range(60): fn(H[0][1], Y)
        for r in range(7):
            for n, L in H:
                for _ in range(10): fn(L, Y)
                torch.cuda.synchronize()
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(30): fn(L, Y)
                e1.record(); torch.cuda.synchronize()
                res[n].append(e0.elapsed_time(e1) / 30 * 1e3)
        md = {n: float(np.median(v)) for n, v in res.items()}
        print(f"M{M} N{N} K{K} {tag:8s}: direct {md['direct']:7.1f} us   lds {md['lds']:7.1f} us   lds / direct {md['lds'] / md['direct']:.3f}", flush=True)
