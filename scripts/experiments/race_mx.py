import sys, os; sys.path.insert(0,".")
import torch, msq
from msq import qlinear
dev=torch.device("cuda:0"); torch.manual_seed(2)
for (M,N,K) in [(640,16384,1024),(640,16384,4096),(640,16384,512),(2048,5120,1024),(704,16384,1024),(2560,4096,1024),(640,16384,256)]:
    W=torch.randn(N,K,device=dev)*0.02
    P=qlinear.mx_pack_weight(W)
    X=torch.randn(M,K,device=dev); xp=qlinear.mx_pack_act(X)
    y0=qlinear.qlinear_mx_w4a8(xp,P,None,torch.float32)
    d=torch.zeros((),dtype=torch.int64,device=dev); nbad=0
    for _ in range(600):
        y=qlinear.qlinear_mx_w4a8(xp,P,None,torch.float32)
        d+=(y!=y0).any().to(torch.int64)
    print(os.environ.get("MSQ_MX_MF","auto"),M,N,K,int(d.item()),"differing of 600", flush=True)
