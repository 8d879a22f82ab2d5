"""MX GEMM with two k-groups per block (auto rule) vs MSQ_MX_KG=1, each arm in a child process"""
import os, subprocess, sys
CHILD = r'''
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(sys.argv[1])))))
import torch, msq
from msq import qlinear, quant
from msq._lib import lib, ptr, check, current_stream
dev = torch.device("cuda:0"); torch.manual_seed(0)
def t(fn, n=50, warm=120):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n
out = []
for (M, N, K) in [tuple(int(v) for v in a.split(",")) for a in (os.environ.get("SHAPES") or "2048,4096,4096 2048,4096,11008 1536,4096,4096 1024,8192,8192 2048,16384,4096 2048,11008,4096").split()]:
    W = torch.randn(N, K, device=dev) * 0.02
    X = torch.randn(M, K, device=dev)
    xc, xs = qlinear.mx_pack_act(X)
    y = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    r = []
    for wf in ("e2m1", "e3m2", "e4m3"):
        P = qlinear.mx_pack_values(quant.outlier_fakequant(W, 8, 8, "fp4_e2m1", "fp8_e4m3", 2, -1, 32)["out"]) if wf == "e4m3" else qlinear.mx_pack_weight(W, w_fmt=wf)
        us = min(t(lambda: qlinear.qlinear_mx_w4a8((xc, xs), P, None, torch.bfloat16, out=y)) for _ in range(3)) * 1e3
        r.append("%s %.1f us %.0f TF" % (wf, us, 2.0 * M * N * K / us / 1e6))
    out.append("M%d N%d K%d: %s" % (M, N, K, " | ".join(r)))
print("RESULT " + " ;; ".join(out))
'''
ARMS = {"auto": {}, "kg1 mf8": {"MSQ_MX_KG": "1", "MSQ_MX_MF": "8"}, "mf4 ks1": {"MSQ_MX_MF": "4", "MSQ_MX_GEMM_KS": "1"}}
for label, env_add in [(a, ARMS[a]) for a in (os.environ.get("ARMS") or "auto,kg1 mf8").split(",")]:
    env = dict(os.environ); env.update(env_add)
    out = subprocess.run([sys.executable, "-c", CHILD, os.path.abspath(__file__)], env=env, capture_output=True, text=True, timeout=900)
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT")]
    print("[%s]" % label)
    print("\n".join("   " + s for s in line[0][7:].split(" ;; ")) if line else out.stderr[-800:], flush=True)
