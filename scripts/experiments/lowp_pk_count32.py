import os, sys
sys.path.insert(0, "/root/repo/scripts/experiments"); sys.path.insert(0, "/root/repo")
import torch
import lowp_pk_fuzz as F
torch.manual_seed(0)
W = torch.randn(16384, 4096, device=F.dev) * 0.02
W[torch.rand(16384, 4096, device=F.dev) < 0.005] *= 16
import time
for dt in (torch.float16, torch.bfloat16):
    x = W.to(dt)
    for fi, fo, axis, bs in (("int2", "fp4", 0, 16), ("fp4_e2m1", "fp8_e4m3", 1, 32), ("fp4_e2m1", "fp8_e4m3", 0, 32), ("int2", "fp4", 1, 16)):
        F.HANDED.clear()
        F.run(x, fi, fo, 2.0, axis, bs, 8, 1, ("c", "c"), native=False)
        h, w = F.HANDED[("c", "c")]
        # time
        import msq
        msq.quant.CHECK_NAN = False
        f = lambda: msq.quant.outlier_fakequant(x, 8, 8, fi, fo, 2, axis, bs, compute_dtype="float32")
        for _ in range(10): f()
        torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        print(str(dt)[6:], fi, fo, "axis", axis, "bs", bs, ": handed back %d of %d waves (%.2f %%), %.1f us" % (h, w, 100.0 * h / w, e0.elapsed_time(e1) / 20 * 1e3))
