"""act_quant / pack timing vs size (separates host launch overhead from kernel throughput); run under rocprofv3 --kernel-trace --stats for kernel durations"""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import msq
from msq import qlinear
dev = torch.device("cuda:0")
def t(fn, n=20, warm=10):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
K = 4096
for M in (2048, 8192, 32768):
    X = torch.randn(M, K, device=dev)
    for variant, sd in ((0, 2), (1, 5)):
        ms = t(lambda: qlinear.act_quant(X, 8, 8, "fp8_e4m3", "fp8_e4m3", sd, 32, "nearest", False, variant))
        print(f"act_quant v{variant} M{M}: {ms*1e3:7.1f} us {M*K*6/ms/1e6:6.0f} GB/s")
    ms = t(lambda: qlinear.mx_pack_act(X))
    print(f"mx_pack_act M{M}: {ms*1e3:7.1f} us {M*K*(5+1/32)/ms/1e6:6.0f} GB/s")
