# split-K sweep of the bf16-activation fused GEMM (MSQ-U1, fp8 outliers) at mid-size M (MSQ_GEMM_KS forced per child process), HIP-graph replay
import os, subprocess, sys
CHILD = r'''
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(sys.argv[1])))))
import torch, msq
from msq import qlinear
from msq._lib import lib, ptr, check, current_stream
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
for (N, K) in [(4096, 4096), (4096, 11008), (11008, 4096), (16384, 4096)]:
    P = qlinear.pack_weight(torch.randn(N, K, device=dev) * 0.02, 8, 8, "fp4_e2m1", "fp8_e4m3", 2, 32, layout="unified")
    for M in (65, 128, 256, 512):
        X = torch.randn(M, K, device=dev).to(torch.bfloat16)
        y = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        ws = torch.empty(64 * M * N * 4, dtype=torch.uint8, device=dev)
        f = lambda: check(lib().msq_qlinear_bf16(ptr(X), None, ptr(P.out), ptr(P.scl), None, ptr(y), 2, M, N, K, 32, P.in_kind, P.out_kind, ptr(ws), ws.numel(), current_stream(dev)), "g")
        out.append("%.1f" % (graphed(f) * 1e3))
print("RESULT " + " ".join(out))
'''
print("columns: (N,K) in [(4096,4096),(4096,11008),(11008,4096),(16384,4096)] x M in (65,128,256,512), us")
for ks in ("0", "1", "2", "4", "8", "16", "32"):
    env = dict(os.environ)
    if ks != "0": env["MSQ_GEMM_KS"] = ks
    out = subprocess.run([sys.executable, "-c", CHILD, os.path.abspath(__file__)], env=env, capture_output=True, text=True, timeout=600)
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT")]
    print(f"KS={ks if ks != '0' else 'auto':5s}: {line[0][7:] if line else out.stderr[-300:]}", flush=True)
