"""Round 5: run-to-run identity and correctness of the persistent stream-K kernel (k_qgemm256p, MSQ_GEMM_256=3) on shapes whose plan CUTS
tiles (partial tiles handed from block to block through the workspace: sc1 stores, flag, sc1 loads) and on whole-round shapes (stores of one
tile in flight under the next tile's first K-step).  Every second launch runs beside a bandwidth hog on another stream (uneven load: the
condition under which a missing drain / a stale line shows, guide G16); between launches the allocator is churned so that the workspace
changes address and content.  REPS launches per case must equal the first one bit for bit; the first one must equal the dense product of
the operands within 2e-5 max|y| (fp32 out)."""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
os.environ["MSQ_GEMM_256"] = "3"
import msq
from msq import qlinear
dev = torch.device("cuda:0"); torch.manual_seed(7)
REPS = int(os.environ.get("REPS", 300))
CASES = [(2048, 12288, 4096), (2048, 4096, 4096), (2048, 22016, 4096), (2048, 4096, 11008), (600, 768, 2048), (256, 256, 8192), (2048, 16384, 4096), (777, 16384, 1152)]
hog_a = torch.empty(64 << 20, dtype=torch.float32, device=dev); hog_b = torch.empty_like(hog_a)
side = torch.cuda.Stream()
bad = 0
for (M, N, K) in CASES:
    W = torch.randn(N, K, device=dev) * 0.02; W[torch.rand(N, K, device=dev) < 0.005] *= 16
    Xb = torch.randn(M, K, device=dev).to(torch.bfloat16)
    for fo in ("posit8_es1", "fp8_e4m3"):
        P = qlinear.pack_weight(W, 8, 8, "fp4_e2m1", fo, 2, 32, layout="unified")
        ref = Xb.float() @ qlinear.unpack_weight(P, torch.float32).t()
        for ydt in (torch.float32, torch.bfloat16):
            y0 = qlinear.qlinear(Xb, P, None, ydt)
            err = (y0.float() - ref).abs().max().item(); tol = (2e-5 if ydt == torch.float32 else 8e-3) * ref.abs().max().item()
            d = torch.zeros((), dtype=torch.int64, device=dev)
            junk = []
            for r in range(REPS):
                if r % 2:
                    with torch.cuda.stream(side):
                        hog_b.copy_(hog_a)
                if r % 7 == 0:
                    junk.append(torch.randn((1 + r % 5) << 18, device=dev))     # churn: the next workspace lands elsewhere / on other bytes
                    if len(junk) > 3: junk.pop(0)
                d += (qlinear.qlinear(Xb, P, None, ydt) != y0).any().to(torch.int64)
            torch.cuda.synchronize()
            n = int(d.item()); bad += n + (err > tol)
            print("k_qgemm256p M%5d N%6d K%6d %-11s %-8s: max err %.2e (tol %.2e)%s, %d launches, %d differing" % (M, N, K, fo, str(ydt)[6:], err, tol, " !!" if err > tol else "", REPS, n), flush=True)
        del P, ref
print("TOTAL bad:", bad)
sys.exit(1 if bad else 0)
