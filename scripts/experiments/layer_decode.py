# The seven Linears of one Llama-2-7B decoder layer at decode sizes, replayed from a HIP graph:
# q/k/v (one shared activation pack), o, gate/up (shared pack), down.  Paths: MX-FP4 weights, MicroScopiQ e4m3
# operand, MSQ-U1 with bf16 activations, hipBLASLt bf16 on the unpacked weight.  (Attention, norms, SiLU not included.)
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, msq
from msq import qlinear, quant
dev = torch.device("cuda:0"); torch.manual_seed(0)
H, I = 4096, 11008
def t(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n
def graphed(fn, reps=10):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): fn()
        s.synchronize()
        gph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gph, stream=s):
            for _ in range(reps): fn()
    torch.cuda.synchronize()
    return min(t(gph.replay) for _ in range(3)) / reps
shapes = dict(q=(H, H), k=(H, H), v=(H, H), o=(H, H), gate=(I, H), up=(I, H), down=(H, I))
Wq = {n: quant.outlier_fakequant(torch.randn(N, K, device=dev) * 0.02, 8, 8, "fp4_e2m1", "fp8_e4m3", 2, -1, 32)["out"] for n, (N, K) in shapes.items()}
P4 = {n: qlinear.mx_pack_weight(w) for n, w in Wq.items()}
P8 = {n: qlinear.mx_pack_values(w) for n, w in Wq.items()}
PU = {n: qlinear.pack_values(w) for n, w in Wq.items()}
WB = {n: w.to(torch.bfloat16) for n, w in Wq.items()}
# q/k/v and gate/up concatenated along out_features (blocks run along in_features: the packed values are unchanged)
cat = dict(qkv=torch.cat([Wq["q"], Wq["k"], Wq["v"]]), gateup=torch.cat([Wq["gate"], Wq["up"]]))
P4c = {n: qlinear.mx_pack_weight(w) for n, w in cat.items()}
P8c = {n: qlinear.mx_pack_values(w) for n, w in cat.items()}
PUc = {n: qlinear.pack_values(w) for n, w in cat.items()}
WBc = {n: w.to(torch.bfloat16) for n, w in cat.items()}
for M in (1, 16, 32):
    x = torch.randn(M, H, device=dev).to(torch.bfloat16); xi = torch.randn(M, I, device=dev).to(torch.bfloat16)
    def mx(P):
        def f():
            a = qlinear.mx_pack_act(x)
            for n in ("q", "k", "v"): qlinear.qlinear_mx_w4a8(a, P[n])
            qlinear.qlinear_mx_w4a8(x, P["o"])
            b = qlinear.mx_pack_act(x)
            for n in ("gate", "up"): qlinear.qlinear_mx_w4a8(b, P[n])
            qlinear.qlinear_mx_w4a8(xi, P["down"])
        return f
    def u1():
        for n in ("q", "k", "v", "o", "gate", "up"): qlinear.qlinear(x, PU[n])
        qlinear.qlinear(xi, PU["down"])
    def bl():
        for n in ("q", "k", "v", "o", "gate", "up"): x @ WB[n].t()
        xi @ WB["down"].t()
    def mxc(P, Pc):
        def f():
            qlinear.qlinear_mx_w4a8(x, Pc["qkv"]); qlinear.qlinear_mx_w4a8(x, P["o"])
            qlinear.qlinear_mx_w4a8(x, Pc["gateup"]); qlinear.qlinear_mx_w4a8(xi, P["down"])
        return f
    def u1c():
        qlinear.qlinear(x, PUc["qkv"]); qlinear.qlinear(x, PU["o"]); qlinear.qlinear(x, PUc["gateup"]); qlinear.qlinear(xi, PU["down"])
    def blc():
        x @ WBc["qkv"].t(); x @ WB["o"].t(); x @ WBc["gateup"].t(); xi @ WB["down"].t()
    rc = [graphed(f) * 1e3 for f in (mxc(P4, P4c), mxc(P8, P8c), u1c, blc)]
    print(f"M{M:3d}: q/k/v and gate/up concatenated (4 Linears): MX-FP4 {rc[0]:6.1f} us | MicroScopiQ e4m3 operand {rc[1]:6.1f} us | MSQ-U1 bf16-act {rc[2]:6.1f} us | hipBLASLt bf16 {rc[3]:6.1f} us", flush=True)
    r = [graphed(f) * 1e3 for f in (mx(P4), mx(P8), u1, bl)]
    wb = sum(N * K for N, K in shapes.values())
    print(f"M{M:3d}: seven Linears of one layer: MX-FP4 {r[0]:6.1f} us | MicroScopiQ e4m3 operand {r[1]:6.1f} us | MSQ-U1 bf16-act {r[2]:6.1f} us | "
          f"hipBLASLt bf16 {r[3]:6.1f} us  ({wb/1e6:.0f} M weights: {wb*2/r[3]/1e3:.0f} / {wb*8.25/8/r[2]/1e3:.0f} / {wb*4.25/8/r[0]/1e3:.0f} GB/s bf16 / MSQ-U1 / MX-FP4)", flush=True)
