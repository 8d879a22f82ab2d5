# MX-native decode (M <= 16): k_mxgemv + k_splitk_reduce replayed from a HIP graph, next to the bf16-activation decode
# kernel on the MSQ-U1 weight and hipBLASLt bf16 on the unpacked weight
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, msq
from msq import qlinear
from msq._lib import lib, ptr, check, current_stream
dev = torch.device("cuda:0"); torch.manual_seed(0)
def t(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n
def graphed(fn, reps=20):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): fn()
        s.synchronize()
        gph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gph, stream=s):
            for _ in range(reps): fn()
    torch.cuda.synchronize()
    return min(t(gph.replay) for _ in range(3)) / reps
for (N, K) in [(16384, 4096), (4096, 4096), (11008, 4096), (4096, 11008)]:
    W = torch.randn(N, K, device=dev) * 0.02
    Pm = qlinear.mx_pack_weight(W)
    Pu = qlinear.pack_weight(W, 8, 8, "fp4_e2m1", "fp8_e4m3", 2, 32, layout="unified")
    P8 = qlinear.mx_pack_values(qlinear.unpack_weight(Pu))      # the same MicroScopiQ values as an e4m3 MFMA operand
    Wu = qlinear.unpack_weight(Pu, torch.bfloat16)
    for M in (1, 16):
        X = torch.randn(M, K, device=dev); Xb = X.to(torch.bfloat16)
        xc, xs = qlinear.mx_pack_act(X)
        y = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        wsb = lib().msq_qlinear_mx_w4a8_workspace_bytes(M, N, K); ws = torch.empty(max(wsb, 1), dtype=torch.uint8, device=dev)
        def gemm():
            check(lib().msq_qlinear_mx_w4a8(ptr(xc), ptr(xs), ptr(Pm.codes), ptr(Pm.scales), None, ptr(y), 2, M, N, K, ptr(ws), wsb,
                                            current_stream(dev)), "gemm")
        def gemm8():
            check(lib().msq_qlinear_mx_w8a8(ptr(xc), ptr(xs), ptr(P8.codes), ptr(P8.scales), None, ptr(y), 2, M, N, K, ptr(ws), wsb,
                                            current_stream(dev)), "gemm8")
        t8 = graphed(gemm8)
        tg = graphed(gemm); te = graphed(lambda: qlinear.qlinear_mx_w4a8(X, Pm)); tu = graphed(lambda: qlinear.qlinear(Xb, Pu)); tb = graphed(lambda: Xb @ Wu.t())
        wb = N * K * 4.25 / 8
        print(f"N{N:5d} K{K:5d} M{M:2d}: MX GEMM {tg*1e3:5.1f} us ({wb/tg/1e6:5.0f} GB/s) | + act pack {te*1e3:5.1f} us | MicroScopiQ e4m3 operand {t8*1e3:5.1f} us ({N*K*8.25/8/t8/1e6:5.0f} GB/s) | MSQ-U1 bf16-act {tu*1e3:5.1f} us | "
              f"hipBLASLt bf16 {tb*1e3:5.1f} us", flush=True)
