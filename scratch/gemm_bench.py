import sys, os, ctypes as C
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import msq
from msq import qlinear
dev = torch.device("cuda:0")
torch.manual_seed(0)
N,K,M = 16384,4096,2048
outs = sys.argv[1].split(",") if len(sys.argv)>1 else ["fp8_e4m3","posit8_es1"]
reps = int(os.environ.get("REPS","3"))
W = torch.randn(N,K,device=dev)*0.02; W[torch.rand(N,K,device=dev)<0.005]*=16
X = torch.randn(M,K,device=dev).to(torch.bfloat16)
for fo in outs:
    fi = "fp4_e2m1" if fo!="int8" else "int8"
    P = qlinear.pack_weight(W,8,8,fi,fo,2,32)
    Wu = qlinear.unpack_weight(P, torch.bfloat16)
    Y = qlinear.qlinear(X,P); Yr = X @ Wu.t()
    err = (Y.float()-Yr.float()).abs().max().item()
    res=[]
    for r in range(reps):
        torch.cuda.synchronize()
        e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): Y = qlinear.qlinear(X,P)
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1)/50)
    ms=min(res)
    print("variant",os.environ.get("MSQ_GEMM_VARIANT","default"),fi,fo,"maxdiff vs hipblaslt %.2e"%err," %.1f us  %.1f TFLOP/s (median %.1f)"%(ms*1e3, 2*M*N*K/ms/1e9, 2*M*N*K/np.median(res)/1e9))
