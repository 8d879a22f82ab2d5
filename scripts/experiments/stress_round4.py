"""Round 4: run-to-run identity and correctness of the 256-row GEMM kernels (k_qgemm256: both unified layouts; k_mxgemm256: MX-FP4, exact e4m3,
MX-FP6 operands) over K-step counts 1, 2, 3, odd, even, ragged M, both output dtypes; half of the repetitions run beside a bandwidth hog on
a second stream (uneven load).  REPS launches per case must equal the first one bit for bit, and the first one must equal the dense product
of the operands (bf16 path: fp32 accumulation, 2e-5 max|y|; MX path: the scaled MFMA's own accumulation against float64, elementwise 2^-11 sum |products|, the bound of the MX tests).
MSQ_GEMM_256 / MSQ_MX_256 = FORM force the kernels onto every shape: FORM=1 (default) the 256-row form, FORM=2 the 128-row form (MF = 8)."""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
FORM = os.environ.get("FORM", "1")
os.environ["MSQ_GEMM_256"] = FORM; os.environ["MSQ_MX_256"] = FORM
import msq
from msq import qlinear, quant
dev = torch.device("cuda:0"); torch.manual_seed(5)
REPS = int(os.environ.get("REPS", 300))
CASES = [(2048, 16384, 4096), (2048, 16384, 128), (2048, 16384, 384), (2048, 16384, 640), (4096, 8192, 1024), (1990, 16384, 1152), (8192, 4096, 512), (300, 2304, 4224)]
hog_a = torch.empty(64 << 20, dtype=torch.float32, device=dev); hog_b = torch.empty_like(hog_a)
side = torch.cuda.Stream()
bad = 0
def hammer(fn, ref, tol, tag, bound=None):
    global bad
    y0 = fn()
    err = (y0.float() - ref).abs().max().item()
    if bound is not None:                                   # elementwise bound (MX path): |y - exact| <= 2^-11 sum |products|
        err = ((y0.float() - ref).abs() / bound).max().item(); tol = 1.0
    d = torch.zeros((), dtype=torch.int64, device=dev)
    for r in range(REPS):
        if r % 2:
            with torch.cuda.stream(side):
                hog_b.copy_(hog_a)
        d += (fn() != y0).any().to(torch.int64)
    torch.cuda.synchronize()
    n = int(d.item()); bad += n + (err > tol)
    print("%s: max err %.2e (tol %.2e)%s, %d launches, %d differing" % (tag, err, tol, " !!" if err > tol else "", REPS, n), flush=True)
for (M, N, K) in CASES:
    W = torch.randn(N, K, device=dev) * 0.02; W[torch.rand(N, K, device=dev) < 0.005] *= 16
    X = torch.randn(M, K, device=dev); Xb = X.to(torch.bfloat16)
    for fo in ("posit8_es1", "fp8_e4m3"):
        P = qlinear.pack_weight(W, 8, 8, "fp4_e2m1", fo, 2, 32, layout="unified")
        ref = Xb.float() @ qlinear.unpack_weight(P, torch.float32).t()
        for ydt in (torch.float32, torch.bfloat16):
            hammer(lambda: qlinear.qlinear(Xb, P, None, ydt), ref, (2e-5 if ydt == torch.float32 else 8e-3) * ref.abs().max().item(),
                   ("k_qgemm256 MF%-2d M%5d N%6d K%6d %-11s %-8s" % (16 // int(FORM), M, N, K, fo, str(ydt)[6:])))
        del P, ref
    xp = qlinear.mx_pack_act(X)
    Xq = msq.mx_ops._quantize_mx(X, 8, "fp8_e4m3", axes=[-1], block_size=32)
    for name in ("fp4", "e4m3", "e3m2"):
        if name == "e4m3":
            Wq = quant.outlier_fakequant(W, 8, 8, "fp4_e2m1", "fp8_e4m3", 2, -1, 32)["out"]; P = qlinear.mx_pack_values(Wq)
        else:
            ef = {"fp4": "fp4_e2m1", "e3m2": "fp6_e3m2"}[name]
            Wq = msq.mx_ops._quantize_mx(W, 8, ef, axes=[-1], block_size=32); P = qlinear.mx_pack_weight(W, w_fmt={"fp4": "e2m1", "e3m2": "e3m2"}[name])
        ref = Xq.double() @ Wq.double().t()
        bound = (Xq.abs() @ Wq.abs().t()) * 2.0 ** -11 + 1e-6
        hammer(lambda: qlinear.qlinear_mx_w4a8(xp, P, None, torch.float32), ref.float(), 1.0, "k_mxgemm256 MF%-2d M%5d N%6d K%6d %-5s f32 (err / bound)" % (16 // int(FORM), M, N, K, name), bound)
        del P, ref, Wq
print("TOTAL bad:", bad)
sys.exit(1 if bad else 0)
