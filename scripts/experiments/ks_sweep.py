import sys, os
sys.path.insert(0, '/root/repo')
import torch, msq
from msq import qlinear
dev = torch.device("cuda:0"); torch.manual_seed(0)
tag = "WM=%s KS=%s" % (os.environ.get("MSQ_GEMM_WM", "auto"), os.environ.get("MSQ_GEMM_KS", "auto"))
for (N,K) in [(4096,4096),(4096,11008),(11008,4096)]:
    if N % 256: N = (N // 256) * 256
    W = torch.randn(N,K,device=dev)*0.02
    P = qlinear.pack_weight(W,8,8,"fp4_e2m1","fp8_e4m3",2,32)
    for M in (128,256,512,1024,2048):
        X = torch.randn(M,K,device=dev).to(torch.bfloat16)
        def t(fn,n=20):
            fn(); torch.cuda.synchronize()
            e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n): fn()
            e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n
        ms=t(lambda: qlinear.qlinear(X,P))
        print(f"{tag} N{N} K{K} M{M:5d}: {ms*1e3:7.1f} us {2*M*N*K/ms/1e9:7.1f} TF", flush=True)
