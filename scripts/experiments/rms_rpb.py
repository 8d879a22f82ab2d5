import sys, os, torch
sys.path.insert(0, "/root/repo")
import msq
from msq import vector_ops as V
from msq._lib import lib
dev = torch.device("cuda:0")
specs = msq.specs.finalize_mx_specs({"w_elem_format": "fp4_e2m1", "a_elem_format": "fp8_e4m3", "scale_bits": 8, "block_size": 32, "custom_cuda": True, "bfloat": 16})
def t(fn, n=20, reps=5):
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
for M in (2048, 8192):
    x = torch.randn(M, 4096, device=dev); w = torch.ones(4096, device=dev); xb = x.to(torch.bfloat16)
    for rpb in (1, 2, 3, 4, 8):
        lib().msq_set_tuning(b"MSQ_RMS_RPB", rpb)
        print("M", M, "rows per block", rpb, "fused f32 %.1f us  bf16 %.1f us   plain %.1f us" % (t(lambda: V.rms_norm_mx_pack(x, w, None, 1e-6, specs)), t(lambda: V.rms_norm_mx_pack(xb, w, None, 1e-6, specs)), t(lambda: V.rms_norm(x, w, None, 1e-6, specs))), flush=True)
