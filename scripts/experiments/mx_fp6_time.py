"""MX GEMM with fp4 / fp6 / e4m3 weight operands (same activations), GEMM only, warm clocks"""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import msq
from msq import qlinear, quant
from msq._lib import lib, ptr, check, current_stream
dev = torch.device("cuda:0"); torch.manual_seed(0)
def t(fn, n=50, warm=120):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n
for (M, N, K) in [(1, 16384, 4096), (16, 16384, 4096), (64, 4096, 4096), (256, 16384, 4096), (2048, 16384, 4096), (8192, 16384, 4096), (2048, 4096, 4096), (8192, 4096, 11008)]:
    W = torch.randn(N, K, device=dev) * 0.02
    X = torch.randn(M, K, device=dev)
    xc, xs = qlinear.mx_pack_act(X)
    r = []
    for wf in ("e2m1", "e3m2", "e4m3"):
        P = qlinear.mx_pack_values(quant.outlier_fakequant(W, 8, 8, "fp4_e2m1", "fp8_e4m3", 2, -1, 32)["out"]) if wf == "e4m3" else qlinear.mx_pack_weight(W, w_fmt=wf)
        y = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        us = min(t(lambda: qlinear.qlinear_mx_w4a8((xc, xs), P, None, torch.bfloat16, out=y)) for _ in range(3)) * 1e3
        r.append("%s %.1f us %.0f TF %.0f GB/s" % (wf, us, 2.0 * M * N * K / us / 1e6, P.nbytes / us / 1e3))
    print("M%d N%d K%d: %s" % (M, N, K, " | ".join(r)), flush=True)
