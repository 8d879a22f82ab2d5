# interleaved A/B of several builds of libmsq_hip.so in ONE process (methodology rule 24)
import sys, os, ctypes as C
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import msq
from msq import qlinear, _lib
dev = torch.device("cuda:0")
torch.manual_seed(0)
libs = sys.argv[1].split(",")
outs = sys.argv[2].split(",") if len(sys.argv) > 2 else ["fp8_e4m3", "posit8_es1"]
M = int(os.environ.get("M", 2048)); N = int(os.environ.get("N", 16384)); K = int(os.environ.get("K", 4096))
rounds = int(os.environ.get("ROUNDS", 5)); iters = int(os.environ.get("ITERS", 30))
handles = []
for p in libs:
    L = C.CDLL(p); fn = L.msq_qlinear_bf16; fn.restype = C.c_int; fn.argtypes = _lib._SIGS["msq_qlinear_bf16"][1]
    handles.append((os.path.basename(p), fn))
W = torch.randn(N, K, device=dev) * 0.02; W[torch.rand(N, K, device=dev) < 0.005] *= 16
X = torch.randn(M, K, device=dev).to(torch.bfloat16)
for fo in outs:
    fi = "fp4_e2m1" if fo != "int8" else "int8"
    P = qlinear.pack_weight(W, 8, 8, fi, fo, 2, 32, layout=os.environ.get("LAYOUT", "unified"))
    Wu = qlinear.unpack_weight(P, torch.bfloat16); Yr = X @ Wu.t()
    Y = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    def call(fn):
        rc = fn(_lib.ptr(X), _lib.ptr(P.inl), _lib.ptr(P.out), _lib.ptr(P.scl), None, _lib.ptr(Y), 2, M, N, K, P.block, P.in_kind, P.out_kind, None, 0, _lib.current_stream())
        assert rc == 0, rc
    res = {n: [] for n, _ in handles}; errs = {}
    for n, fn in handles:
        Y.zero_(); call(fn); torch.cuda.synchronize(); errs[n] = (Y.float() - Yr.float()).abs().max().item()
    for r in range(rounds):
        for n, fn in handles:
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters): call(fn)
            e1.record(); torch.cuda.synchronize()
            res[n].append(e0.elapsed_time(e1) / iters)
    for n, _ in handles:
        t = np.array(res[n]); fl = 2.0 * M * N * K / 1e9
        print("%-28s %-10s err %.1e  min %.1f us = %.0f TF | median %.0f TF" % (n, fo, errs[n], t.min() * 1e3, fl / t.min(), fl / np.median(t)))
