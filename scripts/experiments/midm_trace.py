"""a few launches of the bf16-activation GEMM and the MX GEMM at mid M on 4096 x 4096 for rocprofv3 --kernel-trace"""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import msq
from msq import qlinear, quant
dev = torch.device("cuda:0"); torch.manual_seed(0)
N, K = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4096, 4096)
W = torch.randn(N, K, device=dev) * 0.02
Wq = quant.outlier_fakequant(W, 8, 8, "fp4_e2m1", "fp8_e4m3", 2, -1, 32)["out"]
Pu = qlinear.pack_values(Wq); P4 = qlinear.mx_pack_weight(W)
for M in (64, 65, 128, 256, 512):
    Xb = torch.randn(M, K, device=dev).to(torch.bfloat16); X = Xb.float()
    for _ in range(10):
        qlinear.qlinear(Xb, Pu)
    for _ in range(10):
        qlinear.qlinear_mx_w4a8(X, P4)
    torch.cuda.synchronize()
