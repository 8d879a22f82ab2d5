import sys, os
sys.path.insert(0, '/root/repo')
import torch, msq
from msq import qlinear
dev = torch.device("cuda:0")
for (N,K,fo) in [(4096,11008,"fp8_e4m3"),(4096,4096,"fp8_e4m3"),(11008,4096,"posit8_es1"),(16384,4096,"fp8_e4m3")]:
    g = torch.Generator(device=dev).manual_seed(5)
    W = torch.randn(N, K, generator=g, device=dev) * 0.02
    P = qlinear.pack_weight(W, 8, 8, "fp4_e2m1", fo, 2, 32, layout="unified")
    Wu = qlinear.unpack_weight(P, torch.float32)
    for M in (65, 128, 1000, 2048):
        X = torch.randn(M, K, generator=g, device=dev).to(torch.bfloat16)
        Yr = X.float() @ Wu.t()
        Y0 = qlinear.qlinear(X, P, None, torch.float32)
        bad = []
        for it in range(100):
            Y = qlinear.qlinear(X, P, None, torch.float32)
            if not torch.equal(Y, Y0):
                d = (Y - Yr).abs(); d0 = (Y0 - Yr).abs()
                w = d if d.max() > d0.max() else d0
                idx = (w > 1e-3 * Yr.abs().max()).nonzero()
                bad.append((it, idx.shape[0], idx[:,0].min().item() if idx.numel() else -1, idx[:,0].max().item() if idx.numel() else -1, idx[:,1].min().item() if idx.numel() else -1, idx[:,1].max().item() if idx.numel() else -1, float(w.max())))
        print(N, K, fo, "M", M, "bad", len(bad), bad[:3], flush=True)
