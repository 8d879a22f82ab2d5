import sys, os
sys.path.insert(0, '/root/repo')
import numpy as np, torch, msq
from msq import qlinear
dev = torch.device("cuda:0"); torch.manual_seed(0)
N,K=16384,4096
W = torch.randn(N,K,device=dev)*0.02; W[torch.rand(N,K,device=dev)<0.005]*=16
def t(fn,n=10):
    fn(); torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n
for fo in ("fp8_e4m3","posit8_es1"):
    P = qlinear.pack_weight(W,8,8,"fp4_e2m1",fo,2,32)
    ms = t(lambda: qlinear.pack_weight(W,8,8,"fp4_e2m1",fo,2,32))
    mu = t(lambda: qlinear.unpack_weight(P, torch.bfloat16))
    mf = t(lambda: qlinear.unpack_weight(P, torch.float32))
    print(f"pack {fo}: {ms*1e3:.0f} us ({(N*K*4+P.nbytes)/ms/1e6:.0f} GB/s algorithmic, incl. torch allocs + status sync) | unpack->bf16 {mu*1e3:.0f} us ({(P.nbytes+N*K*2)/mu/1e6:.0f} GB/s) | unpack->f32 {mf*1e3:.0f} us ({(P.nbytes+N*K*4)/mf/1e6:.0f} GB/s)")
