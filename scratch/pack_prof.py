import sys
sys.path.insert(0, '/root/repo')
import torch, msq
from msq import qlinear
dev = torch.device("cuda:0"); torch.manual_seed(0)
N,K=16384,4096
W = torch.randn(N,K,device=dev)*0.02; W[torch.rand(N,K,device=dev)<0.005]*=16
for fo in ("fp8_e4m3","posit8_es1"):
    for _ in range(3):
        P = qlinear.pack_weight(W,8,8,"fp4_e2m1",fo,2,32)
        Wu = qlinear.unpack_weight(P, torch.bfloat16)
        Wq = msq.quant.quantize_mx_outlier_v1(W,8,8,"fp4_e2m1",fo,"max",2,[-1],32)
        Wq0 = msq.quant.quantize_mx_outlier_v1(W,8,8,"int2","fp4","max",2,[0],16)
torch.cuda.synchronize()
