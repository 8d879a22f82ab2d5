import sys, os
sys.path.insert(0, '/root/repo')
import torch, msq
from msq import qlinear
dev = torch.device("cuda:0"); torch.manual_seed(0)
tag = os.environ.get("MSQ_GEMV_MAX_M", "64")
for (N,K) in [(16384,4096),(4096,4096),(11008,4096),(4096,11008)]:
    W = torch.randn(N,K,device=dev)*0.02
    P = qlinear.pack_weight(W,8,8,"fp4_e2m1","fp8_e4m3",2,32,layout="unified")
    for M in (8,16,24,32,48,64):
        X = torch.randn(M,K,device=dev).to(torch.bfloat16)
        def t(fn,n=30):
            fn(); torch.cuda.synchronize()
            e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n): fn()
            e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n
        ms=t(lambda: qlinear.qlinear(X,P))
        print(f"thr={tag} N{N} K{K} M{M:3d}: {ms*1e3:6.1f} us", flush=True)
