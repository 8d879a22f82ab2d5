# kernel-only timing of the pack / unpack / fake-quant kernels: run under rocprofv3 --kernel-trace --stats
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, msq
from msq import qlinear, quant
dev = torch.device("cuda:0"); torch.manual_seed(0)
N, K = 16384, 4096
W = torch.randn(N, K, device=dev) * 0.02; W[torch.rand(N, K, device=dev) < 0.005] *= 16
for fo in ("fp8_e4m3", "posit8_es1"):
    for layout in ("planes", "unified"):
        for _ in range(6):
            P = qlinear.pack_weight(W, 8, 8, "fp4_e2m1", fo, 2, 32, layout=layout)
        for _ in range(6):
            qlinear.unpack_weight(P, torch.bfloat16)
    for _ in range(6):
        quant.quantize_mx_outlier_v1(W, 8, 8, "fp4_e2m1", fo, "max", 2, [-1], 32)
X = torch.randn(2048, 4096, device=dev)
for _ in range(6):
    qlinear.act_quant(X, 8, 8, "fp8_e4m3", "fp8_e4m3", 5, 32, "nearest", False, 1)
torch.cuda.synchronize()
