# 200 launches each of the dispatcher's choice at M = 64, 128 (4096 x 4096) and 256 on the BASELINE weights, for rocprofv3 --kernel-trace --stats (round 6)
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import msq
from msq import qlinear
dev = torch.device("cuda:0"); torch.manual_seed(0)
for (M, N, K) in ((64, 16384, 4096), (256, 16384, 4096), (64, 12288, 4096), (128, 4096, 4096)):
    W = torch.randn(N, K, device=dev) * 0.02; W[torch.rand(N, K, device=dev) < 0.005] *= 16
    P = qlinear.pack_weight(W, 8, 8, "fp4_e2m1", "posit8_es1", 2, 32, layout="unified")
    X = torch.randn(M, K, device=dev).to(torch.bfloat16)
    y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for _ in range(200):
        qlinear.qlinear(X, P, None, torch.bfloat16)
    torch.cuda.synchronize()
    del W, P
