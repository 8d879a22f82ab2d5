"""Round 5: the activation producers in front of the MX Linear (RMSNorm, silu x up; csrc/msq_vec.hip) -- unfused (producer writes float32,
msq_mx_pack_a8 reads it back) against fused (one launch, the MX-FP8 operand only), Llama-2-7B sizes at M = 2048: hidden 4096 in front of
q/k/v and gate/up, intermediate 11008 in front of down_proj; and the step producer + GEMM.  Device times: 20 calls per HIP graph, median of 5 replays."""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import msq
from msq import qlinear, vector_ops as V
dev = torch.device("cuda:0")
specs = msq.specs.finalize_mx_specs({"w_elem_format": "fp4_e2m1", "a_elem_format": "fp8_e4m3", "scale_bits": 8, "block_size": 32, "custom_cuda": True, "bfloat": 16})
def t(fn, n=20, reps=5):
    """device time per call: n calls captured in a HIP graph (the Python wrappers' host time, ~50 us a call, stays out), median of reps replays"""
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    out = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / n * 1e3)
    return sorted(out)[len(out) // 2]
M, H, I = 2048, 4096, 11008
x = torch.randn(M, H, device=dev); w = torch.ones(H, device=dev)
gu = torch.randn(M, 2 * I, device=dev); gate, up = gu[:, :I], gu[:, I:]
rows = []
a = t(lambda: V.rms_norm(x, w, None, 1e-6, specs)); y = V.rms_norm(x, w, None, 1e-6, specs)
b = t(lambda: qlinear.mx_pack_act(y)); c = t(lambda: V.rms_norm_mx_pack(x, w, None, 1e-6, specs))
gb = lambda us, byts: byts / us / 1e3
print("RMSNorm [%d, %d] f32:   producer %.1f us + packer %.1f us = %.1f us   | fused %.1f us  (%.0f GB/s of 5 B / element, %.2f of 8 TB/s)   x%.2f" %
      (M, H, a, b, a + b, c, gb(c, M * H * 5.03), gb(c, M * H * 5.03) / 8000, (a + b) / c), flush=True)
a = t(lambda: V.silu_mul(gate, up, specs)); y2 = V.silu_mul(gate, up, specs)
a3 = t(lambda: V.simd_mul(V.silu(gate, mx_specs=specs), up, mx_specs=specs))
b = t(lambda: qlinear.mx_pack_act(y2)); c = t(lambda: V.silu_mul(gate, up, specs, pack=True))
print("silu x up [%d, %d] f32: producer %.1f us (silu, then simd_mul: %.1f us) + packer %.1f us = %.1f us   | fused %.1f us  (%.0f GB/s of 9 B / element, %.2f of 8 TB/s)   x%.2f" %
      (M, I, a, a3, b, a + b, c, gb(c, M * I * 9.03), gb(c, M * I * 9.03) / 8000, (a + b) / c), flush=True)
# 16-bit activations read as they are (a bf16 model; the GEMM in front writes bf16): no cast pass, half the bytes in
xb = x.to(torch.bfloat16); gub = gu.to(torch.bfloat16); gb_, ub_ = gub[:, :I], gub[:, I:]
c = t(lambda: V.rms_norm_mx_pack(xb, w, None, 1e-6, specs)); u = t(lambda: qlinear.mx_pack_act(V.rms_norm(xb.float(), w, None, 1e-6, specs)))
print("RMSNorm [%d, %d] bf16:  cast + producer + packer %.1f us | fused, reading bf16 %.1f us  (%.0f GB/s of 3 B / element, %.2f of 8 TB/s)   x%.2f" %
      (M, H, u, c, gb(c, M * H * 3.03), gb(c, M * H * 3.03) / 8000, u / c), flush=True)
c = t(lambda: V.silu_mul(gb_, ub_, specs, pack=True)); u = t(lambda: qlinear.mx_pack_act(V.silu_mul(gb_.float(), ub_.float(), specs)))
print("silu x up [%d, %d] bf16: casts + producer + packer %.1f us | fused, reading bf16 %.1f us  (%.0f GB/s of 5 B / element, %.2f of 8 TB/s)   x%.2f" %
      (M, I, u, c, gb(c, M * I * 5.03), gb(c, M * I * 5.03) / 8000, u / c), flush=True)
# steps: producer + GEMM (q/k/v N = 12288 on the RMSNorm output; down_proj N = 4096 on silu x up)
for name, N, K, unf, fus in (("RMSNorm -> q/k/v", 12288, H, lambda: V.rms_norm(x, w, None, 1e-6, specs), lambda: V.rms_norm_mx_pack(x, w, None, 1e-6, specs)),
                             ("silu x up -> down_proj", 4096, I, lambda: V.silu_mul(gate, up, specs), lambda: V.silu_mul(gate, up, specs, pack=True))):
    for wf in ("e2m1", "e4m3"):
        Wt = torch.randn(N, K, device=dev) * 0.02
        P = qlinear.mx_pack_weight(Wt, w_fmt="e2m1") if wf == "e2m1" else qlinear.mx_pack_values(qlinear.unpack_weight(qlinear.pack_weight(Wt, 8, 8, "fp4_e2m1", "fp8_e4m3", 2, 32, layout="unified")))
        u = t(lambda: qlinear.qlinear_mx_w4a8(unf(), P, None, torch.bfloat16))
        f = t(lambda: qlinear.qlinear_mx_w4a8(fus(), P, None, torch.bfloat16))
        pk = fus(); g_ = t(lambda: qlinear.qlinear_mx_w4a8(pk, P, None, torch.bfloat16))
        fl = 2.0 * M * N * K
        print("%-24s weight operand %s: producer + packer + GEMM %.1f us | fused producer + GEMM %.1f us (GEMM alone %.1f us = %.3f of 5 PF)   x%.3f" %
              (name, wf, u, f, g_, fl / g_ / 1e6 / 5000, u / f), flush=True)
