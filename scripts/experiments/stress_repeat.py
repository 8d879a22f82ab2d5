# Run-to-run identity stress of the kernels added late in round 1: every output must equal the first one, bit for bit,
# over many launches (LDS-DMA tiles, LDS reductions and split-K partials cross waves: this is the race detector).
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, msq
from msq import qlinear, quant
dev = torch.device("cuda:0"); torch.manual_seed(0)
REPS = int(os.environ.get("REPS", 1500))
bad = 0
for (N, K) in [(16384, 4096), (4096, 4096), (4096, 11008)]:
    W = torch.randn(N, K, device=dev) * 0.02; W[torch.rand(N, K, device=dev) < 0.005] *= 16
    Wq = quant.outlier_fakequant(W, 8, 8, "fp4_e2m1", "fp8_e4m3", 2, -1, 32)["out"]
    P4, P8, PU = qlinear.mx_pack_weight(W), qlinear.mx_pack_values(Wq), qlinear.pack_values(Wq)
    for M in (1, 16, 32, 48, 130, 2048):
        X = torch.randn(M, K, device=dev); Xb = X.to(torch.bfloat16)
        xp = qlinear.mx_pack_act(X)
        for name, fn in (("mx fp4", lambda: qlinear.qlinear_mx_w4a8(xp, P4, None, torch.float32)),
                         ("mx e4m3", lambda: qlinear.qlinear_mx_w4a8(xp, P8, None, torch.float32)),
                         ("bf16 act", lambda: qlinear.qlinear(Xb, PU, None, torch.float32))):
            y0 = fn()
            reps = REPS if M <= 130 else REPS // 5
            diff = torch.zeros((), dtype=torch.int64, device=dev)
            for _ in range(reps):
                diff += (fn() != y0).any().to(torch.int64)
            d = int(diff.item()); bad += d
            print(f"N{N:5d} K{K:5d} M{M:4d} {name:8s}: {reps} launches, {d} differing", flush=True)
print("TOTAL differing launches:", bad)
sys.exit(1 if bad else 0)
