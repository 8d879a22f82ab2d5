# Is the MX GEMM clock / power limited?  Same kernel, same shapes, operands with less switching activity:
# random codes vs all-zero activation codes vs all-zero weights and activations.
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, msq
from msq import qlinear
from msq._lib import lib, ptr, check, current_stream
dev = torch.device("cuda:0"); torch.manual_seed(0)
def t(fn, n=50, warm=100):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n
M, N, K = 2048, 16384, 4096
W = torch.randn(N, K, device=dev) * 0.02; W[torch.rand(N, K, device=dev) < 0.005] *= 16
P = qlinear.mx_pack_weight(W); Pz = qlinear.mx_pack_weight(torch.zeros_like(W))
X = torch.randn(M, K, device=dev)
y = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
for name, Pw, Xs in (("random W, random X", P, X), ("random W, zero X", P, torch.zeros_like(X)), ("zero W, zero X", Pz, torch.zeros_like(X))):
    xc, xs = qlinear.mx_pack_act(Xs)
    f = lambda: check(lib().msq_qlinear_mx_w4a8(ptr(xc), ptr(xs), ptr(Pw.codes), ptr(Pw.scales), None, ptr(y), 2, M, N, K, None, 0, current_stream(dev)), "g")
    us = min(t(f) for _ in range(3)) * 1e3
    print(f"{name:22s}: {us:6.1f} us  {2*M*N*K/us/1e6:7.1f} TFLOP/s", flush=True)
