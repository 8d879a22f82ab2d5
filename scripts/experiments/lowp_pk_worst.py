import sys
sys.path.insert(0, "/root/repo/scripts/experiments"); sys.path.insert(0, "/root/repo")
import torch, msq
import lowp_pk_fuzz as F
msq.quant.CHECK_NAN = False
L = F.L
W = torch.randn(16384, 4096, device=F.dev)
def tm(f):
    for _ in range(5): f()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / 10 * 1e3
for scale, dt in ((3e-5, torch.float16), (1e-20, torch.bfloat16), (200.0, torch.float16)):
    x = (W * scale).to(dt)
    for native, cd in ((True, "input"), (False, "float32")):
        for axis, bs, fi, fo in ((-1, 32, "fp4_e2m1", "fp8_e4m3"), (0, 16, "int2", "fp4")):
            F.HANDED.clear()
            F.run(x, fi, fo, 2.0, axis, bs, 8, 1, ("c", "c"), native=native)
            h, w = F.HANDED[("c", "c")]
            f = lambda: msq.quant.outlier_fakequant(x, 8, 8, fi, fo, 2, axis, bs, compute_dtype=cd)
            t1 = tm(f)
            L.msq_set_tuning(b"MSQ_OUTLIER_LOWP_PK", 0); t0 = tm(f); L.msq_set_tuning(b"MSQ_OUTLIER_LOWP_PK", 1)
            print("scale %g %s %s axis %d bs %d %s/%s: handed back %.1f %%, packed route %.1f us, old kernels %.1f us" % (scale, str(dt)[6:], "in-dtype" if native else "float32-sem", axis, bs, fi, fo, 100.0 * h / w, t1, t0))
