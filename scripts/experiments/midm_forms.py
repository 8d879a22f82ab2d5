# Mid-M (64 <= M <= 512) landscape of the fused GEMM: every existing form forced, device time from HIP-graph replays (round 6, item 1).
#   python scripts/experiments/midm_forms.py [M,N,K ...]
import os, subprocess, sys
CHILD = r'''
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(sys.argv[1])))))
import torch, msq
from msq import qlinear
dev = torch.device("cuda:0"); torch.manual_seed(0)
def tg(fn, reps=10, n=20):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for _ in range(reps): fn()
    for _ in range(5): g.replay()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n / reps * 1e3
out = []
for shp in sys.argv[2:]:
    M, N, K = (int(v) for v in shp.split(","))
    W = torch.randn(N, K, device=dev) * 0.02; W[torch.rand(N, K, device=dev) < 0.005] *= 16
    X = torch.randn(M, K, device=dev).to(torch.bfloat16)
    r = []
    ncopy = int(os.environ.get("MIDM_COPIES", "1"))       # > 1: the graph walks `ncopy` copies of the packed weight (cold: 5 x 78 MB > the 256 MB Infinity Cache)
    for fo in ("posit8_es1", "fp8_e4m3"):
        P = qlinear.pack_weight(W, 8, 8, "fp4_e2m1", fo, 2, 32, layout="unified")
        try:
            if ncopy > 1:
                import copy
                Ps = [P]
                for _ in range(ncopy - 1):
                    Q = copy.copy(P); Q.out = P.out.clone(); Q.scl = P.scl.clone(); Q.inl = P.inl.clone() if P.inl is not None else None
                    Ps.append(Q)
                st = torch.cuda.Stream()
                with torch.cuda.stream(st):
                    for Q in Ps: qlinear.qlinear(X, Q, None, torch.bfloat16)
                    torch.cuda.synchronize()
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, stream=st):
                        for _ in range(2):
                            for Q in Ps: qlinear.qlinear(X, Q, None, torch.bfloat16)
                for _ in range(3): g.replay()
                torch.cuda.synchronize()
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10): g.replay()
                e1.record(); torch.cuda.synchronize()
                us = e0.elapsed_time(e1) / 10 / (2 * ncopy) * 1e3
                del Ps
            else:
                us = min(tg(lambda: qlinear.qlinear(X, P, None, torch.bfloat16)) for _ in range(2))
            r.append("%6.1f" % us)
        except Exception as e:
            r.append("  fail " + str(e)[:80])
    if os.environ.get("MIDM_BLAS"):
        Wb = qlinear.unpack_weight(P).to(torch.bfloat16)
        r.append("hipBLASLt %6.1f" % min(tg(lambda: torch.nn.functional.linear(X, Wb)) for _ in range(2)))
    out.append("%s: %s" % (shp, " / ".join(r)))
    del W, X, P
print("RESULT " + " ;; ".join(out))
'''
shapes = sys.argv[1:] or ["64,16384,4096", "128,16384,4096", "256,16384,4096", "512,16384,4096"]
arms = [("default", "MIDM_BLAS=1"), ("qgemm3 ks1", "MSQ_GEMM_256=0 MSQ_GEMM_KS=1"), ("qgemm3 ks2", "MSQ_GEMM_256=0 MSQ_GEMM_KS=2"),
        ("qgemm3 ks4", "MSQ_GEMM_256=0 MSQ_GEMM_KS=4"), ("qgemm3 ks8", "MSQ_GEMM_256=0 MSQ_GEMM_KS=8"),
        ("qgemm3 mf4 ks1", "MSQ_GEMM_256=0 MSQ_GEMM_MF=4 MSQ_GEMM_KS=1"), ("q256 mf16", "MSQ_GEMM_256=1"), ("q256 mf8", "MSQ_GEMM_256=2"),
        ("gemv<=64", "MSQ_GEMV_MAX_M=64"), ("sk form 1 (64-row strip, 8 waves)", "MSQ_GEMM_SK=1"), ("sk form 2 (128-row strip, 4 waves)", "MSQ_GEMM_SK=2"),
        ("sk form 3 (128 x 128, 2 x 2 waves)", "MSQ_GEMM_SK=3")]
extra = os.environ.get("MIDM_ARMS")
if extra:
    arms = [tuple(a.split("=", 1)) for a in extra.split(";")]
print("us posit / fp8, device time (HIP graph replays)")
for label, envs in arms:
    env = dict(os.environ)
    for kv in envs.split():
        k, _, v = kv.partition("="); env[k] = v
    out = subprocess.run([sys.executable, "-c", CHILD, os.path.abspath(__file__)] + shapes, env=env, capture_output=True, text=True, timeout=900)
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT")]
    print("[%s]" % label)
    print("\n".join("   " + s for s in line[0][7:].split(" ;; ")) if line else out.stderr[-800:], flush=True)
