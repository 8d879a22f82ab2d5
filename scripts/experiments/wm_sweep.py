import sys, os
sys.path.insert(0, '/root/repo')
import numpy as np, torch, msq
from msq import qlinear
dev = torch.device("cuda:0"); torch.manual_seed(0)
tag = os.environ.get("MSQ_GEMM_WM", "auto")
for (N,K) in [(16384,4096),(4096,4096),(11008,4096),(4096,11008),(12288,4096),(22016,4096)]:
    if N % 256: N = (N // 256) * 256
    W = torch.randn(N,K,device=dev)*0.02; W[torch.rand(N,K,device=dev)<0.005]*=16
    fo = "fp8_e4m3"
    P = qlinear.pack_weight(W,8,8,"fp4_e2m1",fo,2,32)
    Wu = qlinear.unpack_weight(P, torch.bfloat16)
    for M in (128,256,512,1024,2048,4096,8192):
        X = torch.randn(M,K,device=dev).to(torch.bfloat16)
        Y = qlinear.qlinear(X,P); Yr = X @ Wu.t(); torch.cuda.synchronize()
        err=(Y.float()-Yr.float()).abs().max().item()
        def t(fn,n=20):
            fn(); torch.cuda.synchronize()
            e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n): fn()
            e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n
        ms=t(lambda: qlinear.qlinear(X,P))
        print(f"WM={tag} N{N} K{K} M{M:5d}: fused {ms*1e3:7.1f} us {2*M*N*K/ms/1e9:7.1f} TF err {err:.1e}", flush=True)
