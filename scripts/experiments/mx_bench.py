# MX-native W4A8 (fp4 x fp8 scaled MFMA): operand packing and GEMM timings
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, msq
from msq import qlinear
from msq._lib import lib, ptr, check, current_stream
dev = torch.device("cuda:0"); torch.manual_seed(0)
def t(fn, n=30, warm=30):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n
for (N,K) in [(16384,4096),(4096,4096),(11008,4096),(4096,11008)]:
    if K % 128: K = K // 128 * 128
    W = torch.randn(N,K,device=dev)*0.02; W[torch.rand(N,K,device=dev)<0.005]*=16
    w8 = len(sys.argv) > 1 and sys.argv[1] == "w8"      # MicroScopiQ values as e4m3 codes instead of plain MX-FP4
    P = qlinear.mx_pack_values(msq.quant.outlier_fakequant(W, 8, 8, "fp4_e2m1", "fp8_e4m3", 2, -1, 32)["out"]) if w8 else qlinear.mx_pack_weight(W)
    fn = lib().msq_qlinear_mx_w8a8 if w8 else lib().msq_qlinear_mx_w4a8
    for M in (128, 2048, 8192):
        X = torch.randn(M,K,device=dev)
        xc, xs = qlinear.mx_pack_act(X)
        y = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        wsb = lib().msq_qlinear_mx_w4a8_workspace_bytes(M, N, K); ws = torch.empty(max(wsb, 1), dtype=torch.uint8, device=dev)
        def gemm():
            check(fn(ptr(xc), ptr(xs), ptr(P.codes), ptr(P.scales), None, ptr(y), 2, M, N, K, ptr(ws), wsb, current_stream(dev)), "g")
        tg = t(gemm); tp = t(lambda: qlinear.mx_pack_act(X)); te = t(lambda: qlinear.qlinear_mx_w4a8(X, P))
        print(f"N{N} K{K} M{M:5d}: GEMM {tg*1e3:7.1f} us {2*M*N*K/tg/1e9:7.1f} TF | act pack {tp*1e3:6.1f} us | end to end {te*1e3:7.1f} us {2*M*N*K/te/1e9:7.1f} TF", flush=True)
