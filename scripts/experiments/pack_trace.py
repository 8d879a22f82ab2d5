"""pack_weight of W[16384,4096] in the four layouts, a few times each, for rocprofv3 --kernel-trace"""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import msq
from msq import qlinear
dev = torch.device("cuda:0"); torch.manual_seed(0)
N, K = 16384, 4096
W = torch.randn(N, K, device=dev) * 0.02; W[torch.rand(N, K, device=dev) < 0.005] *= 16
for fo in ("fp8_e4m3", "posit8_es1"):
    for layout in ("planes", "unified"):
        for _ in range(5):
            P = qlinear.pack_weight(W, 8, 8, "fp4_e2m1", fo, 2, 32, layout=layout)
        torch.cuda.synchronize()
        print(fo, layout, P.nbytes)
