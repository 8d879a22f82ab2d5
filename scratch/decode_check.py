import sys, os
sys.path.insert(0, '/root/repo')
import torch, msq
from msq import qlinear
dev = torch.device("cuda:0"); torch.manual_seed(0)
def t(fn,n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n
for (N,K) in [(16384,4096),(11008,4096),(4096,11008)]:
    W = torch.randn(N,K,device=dev)*0.02
    for fo in ("fp8_e4m3","posit8_es1"):
        P = qlinear.pack_weight(W,8,8,"fp4_e2m1",fo,2,32,layout="unified")
        for M in (1,8,16):
            X = torch.randn(M,K,device=dev).to(torch.bfloat16)
            r = [t(lambda: qlinear.qlinear(X,P))*1e3 for _ in range(3)]
            print(f"N{N} K{K} {fo} M{M}: " + " ".join(f"{x:.1f}" for x in r) + " us", flush=True)
