"""The Linears of one Llama-2-7B decoder layer at prefill size (M = 2048 and 4096 tokens): fused QKV [12288 x 4096], o [4096 x 4096],
fused gate / up [22016 x 4096], down [4096 x 11008] -- packed MicroScopiQ weights through the fused GEMM (posit8 / fp8 outliers),
the MX-native path, and hipBLASLt bf16 on the unpacked weights; per-layer time and TFLOP/s (GEMMs only, back to back on one stream)."""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import msq
from msq import qlinear, quant
dev = torch.device("cuda:0"); torch.manual_seed(0)
SHAPES = [("qkv", 12288, 4096), ("o", 4096, 4096), ("gate_up", 22016, 4096), ("down", 4096, 11008)]
def t(fn, n=30, warm=60):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n
Ws = {nm: torch.randn(N, K, device=dev) * 0.02 for nm, N, K in SHAPES}
for W in Ws.values(): W[torch.rand_like(W) < 0.005] *= 16
packs = {}
for fo in ("posit8_es1", "fp8_e4m3"):
    packs[fo] = {nm: qlinear.pack_weight(Ws[nm], 8, 8, "fp4_e2m1", fo, 2, 32, layout="unified") for nm, _, _ in SHAPES}
packs["mx fp4"] = {nm: qlinear.mx_pack_weight(Ws[nm]) for nm, _, _ in SHAPES}
packs["mx e4m3"] = {nm: qlinear.mx_pack_values(quant.outlier_fakequant(Ws[nm], 8, 8, "fp4_e2m1", "fp8_e4m3", 2, -1, 32)["out"]) for nm, _, _ in SHAPES}
dense = {nm: qlinear.unpack_weight(packs["fp8_e4m3"][nm], torch.bfloat16) for nm, _, _ in SHAPES}
for M in (2048, 4096):
    X = {K: torch.randn(M, K, device=dev).to(torch.bfloat16) for K in (4096, 11008)}
    Xp = {K: qlinear.mx_pack_act(X[K].float()) for K in (4096, 11008)}
    flops = sum(2.0 * M * N * K for _, N, K in SHAPES)
    rows = []
    for tag in ("posit8_es1", "fp8_e4m3"):
        ms = t(lambda: [qlinear.qlinear(X[K], packs[tag][nm]) for nm, N, K in SHAPES])
        rows.append((tag + " outliers, fused dequant-GEMM", ms))
    for tag in ("mx fp4", "mx e4m3"):
        ms = t(lambda: [qlinear.qlinear_mx_w4a8(Xp[K], packs[tag][nm]) for nm, N, K in SHAPES])
        rows.append((tag + " operand, scaled MFMA (activations pre-packed)", ms))
    ms = t(lambda: [X[K] @ dense[nm].t() for nm, N, K in SHAPES])
    rows.append(("hipBLASLt bf16 on the unpacked weights", ms))
    for name, ms in rows:
        print("M %4d  %-58s %7.1f us per layer  %6.0f TFLOP/s" % (M, name, ms * 1e3, flops / ms / 1e9), flush=True)
