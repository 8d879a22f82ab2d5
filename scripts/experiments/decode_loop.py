# decode (M <= 16) launches for rocprofv3 --kernel-trace --stats: k_qgemv + k_splitk_reduce durations
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, msq
from msq import qlinear
dev = torch.device("cuda:0"); torch.manual_seed(0)
for (N, K) in [(16384, 4096), (4096, 4096), (11008, 4096), (4096, 11008)]:
    W = torch.randn(N, K, device=dev) * 0.02
    for fo in ("fp8_e4m3", "posit8_es1"):
        P = qlinear.pack_weight(W, 8, 8, "fp4_e2m1", fo, 2, 32, layout="unified")
        for M in (1, 16, 32):
            X = torch.randn(M, K, device=dev).to(torch.bfloat16)
            for _ in range(20):
                qlinear.qlinear(X, P)
torch.cuda.synchronize()
# MX-path decode kernels (k_mxgemv<W8, MG> + k_splitk_reduce), fp4 and e4m3 weight operands
from msq import quant
for (N, K) in [(16384, 4096), (4096, 4096), (11008, 4096), (4096, 11008)]:
    W = torch.randn(N, K, device=dev) * 0.02
    P4 = qlinear.mx_pack_weight(W)
    P8 = qlinear.mx_pack_values(quant.outlier_fakequant(W, 8, 8, "fp4_e2m1", "fp8_e4m3", 2, -1, 32)["out"])
    for M in (1, 16, 32):
        X = torch.randn(M, K, device=dev)
        for _ in range(20):
            qlinear.qlinear_mx_w4a8(X, P4); qlinear.qlinear_mx_w4a8(X, P8)
torch.cuda.synchronize()
