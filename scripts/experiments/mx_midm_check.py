# mid-size M (17 .. 256): MX GEMM paths and the bf16-activation kernel replayed from a HIP graph vs hipBLASLt bf16
# (MSQ_GEMV_MAX_M / MSQ_MX_GEMV_MAX_M = 16 or 64 select the decode kernels' upper M: tuning)
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, msq
from msq import qlinear, quant
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
for (N, K) in [(16384, 4096), (4096, 4096), (4096, 11008)]:
    W = torch.randn(N, K, device=dev) * 0.02
    P4 = qlinear.mx_pack_weight(W)
    Wq = quant.outlier_fakequant(W, 8, 8, "fp4_e2m1", "fp8_e4m3", 2, -1, 32)["out"]
    P8 = qlinear.mx_pack_values(Wq)
    Wb = Wq.to(torch.bfloat16)
    Pu = qlinear.pack_values(Wq)                                # MSQ-U1 plane for the bf16-activation kernel
    for M in (16, 17, 32, 33, 48, 64, 65, 128, 256):
        X = torch.randn(M, K, device=dev); Xb = X.to(torch.bfloat16)
        xc, xs = qlinear.mx_pack_act(X)
        y = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        wsb = lib().msq_qlinear_mx_w4a8_workspace_bytes(M, N, K); ws = torch.empty(max(wsb, 1), dtype=torch.uint8, device=dev)
        def mk(fn, P):
            return lambda: check(fn(ptr(xc), ptr(xs), ptr(P.codes), ptr(P.scales), None, ptr(y), 2, M, N, K, ptr(ws), wsb, current_stream(dev)), "g")
        t4 = graphed(mk(lib().msq_qlinear_mx_w4a8, P4)); t8 = graphed(mk(lib().msq_qlinear_mx_w8a8, P8)); tb = graphed(lambda: Xb @ Wb.t()); tu = graphed(lambda: qlinear.qlinear(Xb, Pu))
        print(f"N{N:5d} K{K:5d} M{M:4d}: fp4 weights {t4*1e3:6.1f} us | e4m3 weights {t8*1e3:6.1f} us | MSQ-U1 bf16-act {tu*1e3:6.1f} us | hipBLASLt bf16 {tb*1e3:6.1f} us", flush=True)
