"""host-side cost of one qlinear / QuantLinear.forward call at decode size (the GPU kernel takes ~5 us: the loop is host-bound)"""
import sys, os, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import msq
from msq import qlinear
dev = torch.device("cuda:0"); torch.manual_seed(0)
N, K = 4096, 4096
W = torch.randn(N, K, device=dev) * 0.02
P = qlinear.pack_weight(W, 8, 8, "fp4_e2m1", "fp8_e4m3", 2, 32, layout="unified")
P4 = qlinear.mx_pack_weight(W)
lin = torch.nn.Linear(K, N, bias=False).to(dev).to(torch.bfloat16)
x = torch.randn(1, K, device=dev).to(torch.bfloat16)
xf = x.float()
def wall(fn, n=3000):
    for _ in range(200): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
print("qlinear (bf16 act, MSQ-U1)      : %.1f us per call" % wall(lambda: qlinear.qlinear(x, P)))
y = torch.empty(1, N, dtype=torch.bfloat16, device=dev)
print("qlinear with out=               : %.1f us per call" % wall(lambda: qlinear.qlinear(x, P, out=y)))
print("qlinear_mx_w4a8 (pack + GEMM)   : %.1f us per call" % wall(lambda: qlinear.qlinear_mx_w4a8(xf, P4)))
print("torch nn.Linear bf16 (hipBLASLt): %.1f us per call" % wall(lambda: lin(x)))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(2000): qlinear.qlinear(x, P)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(12)
