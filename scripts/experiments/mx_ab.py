# interleaved A/B of several builds of libmsq_hip.so on the MX matrix path (msq_qlinear_mx_w4a8 / _w8a8) in ONE process
import sys, os, ctypes as C
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import msq
from msq import qlinear, quant, _lib
dev = torch.device("cuda:0")
torch.manual_seed(0)
libs = sys.argv[1].split(",")
fmts = sys.argv[2].split(",") if len(sys.argv) > 2 else ["fp4", "e4m3"]
M = int(os.environ.get("M", 2048)); N = int(os.environ.get("N", 16384)); K = int(os.environ.get("K", 4096))
rounds = int(os.environ.get("ROUNDS", 5)); iters = int(os.environ.get("ITERS", 30))
W = torch.randn(N, K, device=dev) * 0.02; W[torch.rand(N, K, device=dev) < 0.005] *= 16
X = torch.randn(M, K, device=dev)
xc, xs = qlinear.mx_pack_act(X)
Y = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
for fm in fmts:
    P = qlinear.mx_pack_weight(W, w_fmt="e2m1") if fm == "fp4" else qlinear.mx_pack_values(quant.outlier_fakequant(W, 8, 8, "fp4_e2m1", "fp8_e4m3", 2, -1, 32)["out"])
    name = "msq_qlinear_mx_w4a8" if fm == "fp4" else "msq_qlinear_mx_w8a8"
    handles = []
    for p in libs:
        L = C.CDLL(p); fn = getattr(L, name); fn.restype = C.c_int; fn.argtypes = _lib._SIGS[name][1]
        handles.append((os.path.basename(p), fn))
    def call(fn):
        rc = fn(_lib.ptr(xc), _lib.ptr(xs), _lib.ptr(P.codes), _lib.ptr(P.scales), None, _lib.ptr(Y), 2, M, N, K, None, 0, _lib.current_stream())
        assert rc == 0, rc
    res = {n: [] for n, _ in handles}
    for n, fn in handles:
        for _ in range(20): call(fn)
    for r in range(rounds):
        for n, fn in handles:
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters): call(fn)
            e1.record(); torch.cuda.synchronize()
            res[n].append(e0.elapsed_time(e1) / iters)
    for n, _ in handles:
        t = np.array(res[n]); fl = 2.0 * M * N * K / 1e9
        print("%-34s %-5s min %.1f us = %.0f TF | median %.0f TF = %.3f of 5 PF" % (n, fm, t.min() * 1e3, fl / t.min(), fl / np.median(t), fl / np.median(t) / 5000))
