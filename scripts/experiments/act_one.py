"""a few launches of act_quant (variant from argv) on X[2048,4096] for PMC passes"""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import msq
from msq import qlinear
v = int(sys.argv[1]) if len(sys.argv) > 1 else 0
M = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
X = torch.randn(M, 4096, device="cuda:0")
for _ in range(5):
    qlinear.act_quant(X, 8, 8, "fp8_e4m3", "fp8_e4m3", 5 if v else 2, 32, "nearest", False, v)
torch.cuda.synchronize()
