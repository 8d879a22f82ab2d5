"""Round 3: run-to-run identity and correctness of the staggered eight-wave k_qgemm3 (posit / extension-bit layout; fp8 layout as the
unstaggered control) over K-step counts 1, 2, 3, odd, even, ragged M, both output dtypes; half of the repetitions run beside a
bandwidth hog on a second stream (uneven load).  REPS launches per case must equal the first one bit for bit, and the first one must
equal the dense product of the unpacked weight."""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import msq
from msq import qlinear
dev = torch.device("cuda:0"); torch.manual_seed(5)
REPS = int(os.environ.get("REPS", 400))
CASES = [(2048, 16384, 4096), (2048, 16384, 4160), (2048, 16384, 64), (2048, 16384, 128), (2048, 16384, 192), (2048, 16384, 320),
         (4096, 8192, 1024), (1990, 16384, 1088), (8192, 4096, 512)]
hog_a = torch.empty(64 << 20, dtype=torch.float32, device=dev); hog_b = torch.empty_like(hog_a)
side = torch.cuda.Stream()
bad = 0
for (M, N, K) in CASES:
    W = torch.randn(N, K, device=dev) * 0.02; W[torch.rand(N, K, device=dev) < 0.005] *= 16
    Xb = torch.randn(M, K, device=dev).to(torch.bfloat16)
    for fo in ("posit8_es1", "fp8_e4m3"):
        P = qlinear.pack_weight(W, 8, 8, "fp4_e2m1", fo, 2, 32, layout="unified")
        Wu = qlinear.unpack_weight(P, torch.float32)
        ref = Xb.float() @ Wu.t()
        for ydt in (torch.float32, torch.bfloat16):
            fn = lambda: qlinear.qlinear(Xb, P, None, ydt)
            y0 = fn()
            err = (y0.float() - ref).abs().max().item(); tol = (2e-5 if ydt == torch.float32 else 8e-3) * ref.abs().max().item()
            d = torch.zeros((), dtype=torch.int64, device=dev)
            for r in range(REPS):
                if r % 2:
                    with torch.cuda.stream(side):
                        hog_b.copy_(hog_a)
                d += (fn() != y0).any().to(torch.int64)
            torch.cuda.synchronize()
            n = int(d.item()); bad += n + (err > tol)
            print("M%5d N%6d K%6d %-11s %-8s: max err %.2e (tol %.2e)%s, %d launches, %d differing" % (M, N, K, fo, str(ydt)[6:], err, tol, " !!" if err > tol else "", REPS, n), flush=True)
        del P, Wu, ref
print("TOTAL bad:", bad)
sys.exit(1 if bad else 0)
