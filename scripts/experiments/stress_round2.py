"""Run-to-run identity over every launch shape the round-2 rules can pick (eight-wave blocks, four-wave blocks, 64-row tiles, k-groups,
split-K, decode kernels; bf16-activation and MX paths, all weight operand formats): REPS launches per case must equal the first."""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import msq
from msq import qlinear, quant
dev = torch.device("cuda:0"); torch.manual_seed(4)
REPS = int(os.environ.get("REPS", 500))
CASES = [(2048, 16384, 1024), (2048, 12288, 512), (2048, 4096, 1024), (1536, 4096, 512), (2048, 5120, 512), (640, 16384, 256),
         (1000, 4096, 512), (3072, 4096, 1024), (300, 4096, 4096), (130, 11008, 4096), (48, 4096, 4096), (7, 16384, 4096),
         (1024, 4096, 11008), (768, 4096, 4096), (384, 8192, 4096)]
bad = 0
for (M, N, K) in CASES:
    W = torch.randn(N, K, device=dev) * 0.02; W[torch.rand(N, K, device=dev) < 0.005] *= 16
    Wq = quant.outlier_fakequant(W, 8, 8, "fp4_e2m1", "fp8_e4m3", 2, -1, 32)["out"]
    Xb = torch.randn(M, K, device=dev).to(torch.bfloat16); xp = qlinear.mx_pack_act(Xb.float())
    fns = {"bf16 posit": (lambda P=qlinear.pack_weight(W, 8, 8, "fp4_e2m1", "posit8_es1", 2, 32, layout="unified"): qlinear.qlinear(Xb, P, None, torch.float32)),
           "bf16 fp8": (lambda P=qlinear.pack_values(Wq): qlinear.qlinear(Xb, P, None, torch.float32)),
           "mx fp4": (lambda P=qlinear.mx_pack_weight(W): qlinear.qlinear_mx_w4a8(xp, P, None, torch.float32)),
           "mx fp6": (lambda P=qlinear.mx_pack_weight(W, w_fmt="e3m2"): qlinear.qlinear_mx_w4a8(xp, P, None, torch.float32)),
           "mx e4m3": (lambda P=qlinear.mx_pack_values(Wq): qlinear.qlinear_mx_w4a8(xp, P, None, torch.float32))}
    for name, fn in fns.items():
        y0 = fn()
        d = torch.zeros((), dtype=torch.int64, device=dev)
        for _ in range(REPS):
            d += (fn() != y0).any().to(torch.int64)
        n = int(d.item()); bad += n
        print("M%5d N%6d K%6d %-10s: %d launches, %d differing" % (M, N, K, name, REPS, n), flush=True)
print("TOTAL differing launches:", bad)
sys.exit(1 if bad else 0)
