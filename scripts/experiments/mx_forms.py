# MX matrix path: forms of the GEMM on the 7B projections at M = 2048 (device time, HIP-graph replays), activations already packed.
import os, subprocess, sys
CHILD = r'''
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(sys.argv[1])))))
import torch, msq
from msq import qlinear, quant
dev = torch.device("cuda:0"); torch.manual_seed(0)
def tg(fn, reps=10, n=20):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for _ in range(reps): fn()
    for _ in range(8): g.replay()
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
    X = torch.randn(M, K, device=dev)
    xp = qlinear.mx_pack_act(X)
    r = []
    for key, P in (("fp4", qlinear.mx_pack_weight(W, w_fmt="e2m1")), ("e4m3", qlinear.mx_pack_values(quant.outlier_fakequant(W, 8, 8, "fp4_e2m1", "fp8_e4m3", 2, -1, 32)["out"]))):
        us = min(tg(lambda: qlinear.qlinear_mx_w4a8(xp, P, None, torch.bfloat16)) for _ in range(2))
        r.append("%s %6.1f us %.3f" % (key, us, 2.0 * M * N * K / us / 1e6 / 5000.0))
    out.append("%s: %s" % (shp, " | ".join(r)))
print("RESULT " + " ;; ".join(out))
'''
shapes = sys.argv[1:] or ["2048,12288,4096", "2048,4096,4096", "2048,22016,4096", "2048,4096,11008", "2048,16384,4096"]
for label, envs in (("rule", ""), ("k_mxgemm only", "MSQ_MX_256=0"), ("256-row forced", "MSQ_MX_256=1"), ("128-row forced", "MSQ_MX_256=2")):
    env = dict(os.environ)
    for kv in envs.split():
        k, _, v = kv.partition("="); env[k] = v
    o = subprocess.run([sys.executable, "-c", CHILD, os.path.abspath(__file__)] + shapes, env=env, capture_output=True, text=True, timeout=900)
    line = [l for l in o.stdout.splitlines() if l.startswith("RESULT")]
    print("[%s]" % label)
    print("\n".join("   " + s for s in line[0][7:].split(" ;; ")) if line else o.stderr[-600:], flush=True)
