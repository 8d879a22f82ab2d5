# As mx_power_check.py for the bf16-activation fused dequant-GEMM (default bench shape, MSQ-U1 + extension bit):
# random vs all-zero activations / weights, same instruction stream.
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, msq
from msq import qlinear
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
for fo in ("posit8_es1", "fp8_e4m3"):
    P = qlinear.pack_weight(W, 8, 8, "fp4_e2m1", fo, 2, 32, layout="unified")
    Pz = qlinear.pack_weight(torch.zeros_like(W), 8, 8, "fp4_e2m1", fo, 2, 32, layout="unified")
    X = torch.randn(M, K, device=dev).to(torch.bfloat16)
    for name, Pw, Xs in (("random W, random X", P, X), ("random W, zero X", P, torch.zeros_like(X)), ("zero W, zero X", Pz, torch.zeros_like(X))):
        us = min(t(lambda: qlinear.qlinear(Xs, Pw)) for _ in range(3)) * 1e3
        print(f"{fo:11s} {name:22s}: {us:6.1f} us  {2*M*N*K/us/1e6:7.1f} TFLOP/s", flush=True)
