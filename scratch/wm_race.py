import sys, os
sys.path.insert(0, '/root/repo')
import numpy as np, torch, msq
from msq import qlinear
dev = torch.device("cuda:0"); torch.manual_seed(0)
for (N,K) in [(4096,4096),(11008,4096),(22016,4096),(16384,4096)]:
    W = torch.randn(N,K,device=dev)*0.02; W[torch.rand(N,K,device=dev)<0.005]*=16
    P = qlinear.pack_weight(W,8,8,"fp4_e2m1","fp8_e4m3",2,32)
    Wu = qlinear.unpack_weight(P, torch.bfloat16)
    for M in (128,1024,2048):
        X = torch.randn(M,K,device=dev).to(torch.bfloat16)
        Yr = (X @ Wu.t()).float()
        bad = []
        for it in range(30):
            Y = qlinear.qlinear(X,P).float()
            d = (Y-Yr).abs()
            if d.max().item() > 0.5:
                idx = (d > 0.5).nonzero()
                bad.append((it, idx.shape[0], idx[:,0].min().item(), idx[:,0].max().item(), idx[:,1].min().item(), idx[:,1].max().item()))
        print(N,K,M,"bad runs:",len(bad), bad[:4], flush=True)
