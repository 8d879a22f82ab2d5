# k_mxgemm ablation: times the GEMM of every libmsq_hip_mxabl<V>.so in scripts/experiments/abl/ (built by
# build_mx_ablation.sh) next to the product library, fp4 and e4m3 weight operands, M2048 N16384 K4096.
# The parent never touches the GPU; every variant runs in its own child process (MSQ_LIB_OVERRIDE).
import glob, os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
CHILD = r'''
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(sys.argv[1])))))
import torch, msq
from msq import qlinear, quant
from msq._lib import lib, ptr, check, current_stream
dev = torch.device("cuda:0"); torch.manual_seed(0)
def t(fn, n=50, warm=100):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n
M, N, K = 2048, 16384, 4096
W = torch.randn(N, K, device=dev) * 0.02; W[torch.rand(N, K, device=dev) < 0.005] *= 16
X = torch.randn(M, K, device=dev)
xc, xs = qlinear.mx_pack_act(X)
y = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
res = []
for w8 in (False, True):
    P = qlinear.mx_pack_values(quant.outlier_fakequant(W, 8, 8, "fp4_e2m1", "fp8_e4m3", 2, -1, 32)["out"], allow_inexact=True) if w8 else qlinear.mx_pack_weight(W)
    fn = lib().msq_qlinear_mx_w8a8 if w8 else lib().msq_qlinear_mx_w4a8
    def gemm():
        check(fn(ptr(xc), ptr(xs), ptr(P.codes), ptr(P.scales), None, ptr(y), 2, M, N, K, None, 0, current_stream(dev)), "g")
    res.append(min(t(gemm) for _ in range(3)) * 1e3)
print("RESULT %.1f %.1f" % tuple(res))
'''
libs = [("product", None)] + sorted((os.path.basename(p)[len("libmsq_hip_mxabl"):-3], p) for p in glob.glob(os.path.join(HERE, "abl", "libmsq_hip_mxabl*.so")))
for tag, path in libs:
    env = dict(os.environ)
    if path: env["MSQ_LIB_OVERRIDE"] = path
    out = subprocess.run([sys.executable, "-c", CHILD, os.path.abspath(__file__)], env=env, capture_output=True, text=True, timeout=300)
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT")]
    print(f"MSQ_MXABL={tag:8s}: fp4 weights / e4m3 weights GEMM us: {line[0][7:] if line else out.stderr[-300:]}", flush=True)
