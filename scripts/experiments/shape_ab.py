# A/B of the fused GEMM over a list of shapes under environment switches (every arm in its own child process).
#   python scripts/experiments/shape_ab.py "base=" "wn8=MSQ_GEMM_WN=8" [-- M,N,K ...]
import os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
CHILD = r'''
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(sys.argv[1])))))
import torch, msq
from msq import qlinear
dev = torch.device("cuda:0"); torch.manual_seed(0)
def t(fn, n=100, warm=150):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n
out = []
for shp in sys.argv[2:]:
    M, N, K = (int(v) for v in shp.split(","))
    W = torch.randn(N, K, device=dev) * 0.02; W[torch.rand(N, K, device=dev) < 0.005] *= 16
    X = torch.randn(M, K, device=dev).to(torch.bfloat16)
    r = []
    for fo in ("posit8_es1", "fp8_e4m3"):
        P = qlinear.pack_weight(W, 8, 8, "fp4_e2m1", fo, 2, 32, layout="unified")
        us = min(t(lambda: qlinear.qlinear(X, P, None, torch.bfloat16)) for _ in range(3)) * 1e3
        r.append("%.1f us %.0f TF" % (us, 2.0 * M * N * K / us / 1e6))
    out.append("%s: posit %s | fp8 %s" % (shp, r[0], r[1]))
    del W, X, P
print("RESULT " + " ;; ".join(out))
'''
args = sys.argv[1:]
shapes = ["2048,16384,4096", "8192,16384,4096", "1024,16384,4096", "512,16384,4096", "2048,4096,4096", "2048,4096,11008", "2048,8192,3584", "2048,8192,28672"]
if "--" in args:
    i = args.index("--"); shapes = args[i + 1:]; args = args[:i]
for arm in args:
    label, _, envs = arm.partition("=")
    env = dict(os.environ)
    for kv in envs.split():
        k, _, v = kv.partition("="); env[k] = v
    out = subprocess.run([sys.executable, "-c", CHILD, os.path.abspath(__file__)] + shapes, env=env, capture_output=True, text=True, timeout=900)
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT")]
    print("[%s]" % label)
    print("\n".join("   " + s for s in line[0][7:].split(" ;; ")) if line else out.stderr[-500:], flush=True)
