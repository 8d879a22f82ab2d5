import sys, os
sys.path.insert(0, '/root/repo')
import torch, msq
from msq import qlinear
dev = torch.device("cuda:0"); torch.manual_seed(0)
N,K,M=16384,4096,2048
W = torch.randn(N,K,device=dev)*0.02; W[torch.rand(N,K,device=dev)<0.005]*=16
X = torch.randn(M,K,device=dev).to(torch.bfloat16)
for fo in ("fp8_e4m3","posit8_es1"):
    P = qlinear.pack_weight(W,8,8,"fp4_e2m1",fo,2,32,layout="unified")
    for _ in range(20): qlinear.qlinear(X,P)
    torch.cuda.synchronize()
    # back-to-back
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100): qlinear.qlinear(X,P)
    e1.record(); torch.cuda.synchronize(); b2b=e0.elapsed_time(e1)/100
    # isolated: one event pair per launch, device idle before each
    ts=[]
    for _ in range(100):
        torch.cuda.synchronize()
        a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
        a.record(); qlinear.qlinear(X,P); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    ts.sort()
    print(f"{fo}: back-to-back {b2b*1e3:.1f} us/launch ({2*M*N*K/b2b/1e9:.0f} TF); isolated median {ts[50]*1e3:.1f} us min {ts[0]*1e3:.1f} ({2*M*N*K/ts[50]/1e9:.0f} TF)")
