import sys, os
sys.path.insert(0, '/root/repo')
import torch, msq
from msq import qlinear
dev = torch.device("cuda:0")
iters = int(os.environ.get("ITERS", 1500))
for (N,K,fo,layout) in [(4096,11008,"fp8_e4m3","unified"),(4096,11008,"fp8_e4m3","planes"),(4096,4096,"fp8_e4m3","unified"),(11008,4096,"posit8_es1","unified")]:
    g = torch.Generator(device=dev).manual_seed(5)
    W = torch.randn(N, K, generator=g, device=dev) * 0.02
    W[torch.rand(N, K, generator=g, device=dev) < 0.005] *= 16
    P = qlinear.pack_weight(W, 8, 8, "fp4_e2m1", fo, 2, 32, layout=layout)
    for M in (65, 128, 1000, 2048):
        X = torch.randn(M, K, generator=g, device=dev).to(torch.bfloat16)
        Y0 = qlinear.qlinear(X, P, None, torch.float32)
        bad = []
        for it in range(iters):
            Y = qlinear.qlinear(X, P, None, torch.float32)
            if not torch.equal(Y, Y0):
                d = (Y - Y0).abs()
                idx = (d > 0).nonzero()
                bad.append((it, idx.shape[0], idx[:,0].min().item(), idx[:,0].max().item(), idx[:,1].min().item(), idx[:,1].max().item(), float(d.max())))
        print(N, K, fo, layout, "M", M, "bad", len(bad), bad[:4], flush=True)
