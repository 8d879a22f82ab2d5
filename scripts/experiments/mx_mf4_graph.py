"""MX GEMM at mid M from HIP-graph replays (no host time): the shipped launch rules vs forced 64-row blocks in one pass
(MSQ_MX_MF=4 MSQ_MX_GEMM_KS=1), each arm in a child process"""
import os, subprocess, sys
CHILD = r'''
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(sys.argv[1])))))
import torch, msq
from msq import qlinear, quant
dev = torch.device("cuda:0"); torch.manual_seed(0)
def t(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n
def graphed(fn, reps=20):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): fn()
        s.synchronize()
        gph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gph, stream=s):
            for _ in range(reps): fn()
    torch.cuda.synchronize()
    return min(t(gph.replay) for _ in range(3)) / reps
out = []
for (N, K) in [(4096, 4096), (8192, 4096), (11008, 4096), (4096, 11008)]:
    W = torch.randn(N, K, device=dev) * 0.02
    P4 = qlinear.mx_pack_weight(W)
    P8 = qlinear.mx_pack_values(quant.outlier_fakequant(W, 8, 8, "fp4_e2m1", "fp8_e4m3", 2, -1, 32)["out"])
    for M in (128, 256, 384, 512, 768, 1024):
        X = torch.randn(M, K, device=dev); xp = qlinear.mx_pack_act(X)
        y = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        a = graphed(lambda: qlinear.qlinear_mx_w4a8(xp, P4, None, torch.bfloat16, out=y)) * 1e3
        b = graphed(lambda: qlinear.qlinear_mx_w4a8(xp, P8, None, torch.bfloat16, out=y)) * 1e3
        out.append("%d,%d,%d:%.1f/%.1f" % (M, N, K, a, b))
print("RESULT " + " ".join(out))
'''
res = {}
for label, env_add in (("auto", {}), ("mf4 one pass", {"MSQ_MX_MF": "4", "MSQ_MX_GEMM_KS": "1"})):
    env = dict(os.environ); env.update(env_add)
    out = subprocess.run([sys.executable, "-c", CHILD, os.path.abspath(__file__)], env=env, capture_output=True, text=True, timeout=1200)
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT")]
    if not line:
        print(label, out.stderr[-600:]); continue
    for item in line[0][7:].split():
        k, v = item.split(":"); res.setdefault(k, {})[label] = v
print("M,N,K: fp4 / e4m3 operand us   auto | 64-row blocks in one pass")
for k, v in res.items():
    print("%-18s %14s | %14s" % (k, v.get("auto"), v.get("mf4 one pass")))
