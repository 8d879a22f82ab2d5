import sys, os, torch, time
sys.path.insert(0, "/root/repo")
import bench, msq
from msq import qlinear
dev = torch.device("cuda:0")
def tg(fn, n=5):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for seed in (0, 100):
    for layout in ("auto", "unified"):
        W = bench.synth_weight(8192, 28672, dev, seed=seed)
        try:
            P = qlinear.pack_weight(W, 8, 8, "fp4_e2m1", "posit8_es1", 2, 32, layout=layout)
        except Exception as e:
            print("seed", seed, layout, "pack failed:", str(e)[:100]); continue
        del W
        X = torch.randn(2048, 28672, device=dev).to(torch.bfloat16)
        q = qlinear.QuantLinear.from_packed(P, None, out_dtype=torch.bfloat16)
        y = torch.empty(2048, 8192, dtype=torch.bfloat16, device=dev)
        print("seed", seed, layout, "kinds", P.in_kind, P.out_kind, "bits/w %.2f" % P.bits_per_element, "kernel", bench._kernel_name(P, 2048, False),
              "us %.1f" % tg(lambda: q(X, out=y)), flush=True)
        del P, q
