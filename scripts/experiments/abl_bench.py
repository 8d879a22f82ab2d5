# k_qgemm3 ablation: times the fused GEMM (M2048 N16384 K4096, fp4 + posit8 / fp8 outliers) with every
# libmsq_hip_abl<V>.so of scripts/experiments/abl/ (build_abl.sh) next to the product library, for MSQ_GEMM_MF = 8 and 16.
# The weight is packed by the product library in the parent-less child (packing is not ablated).
import glob, os, subprocess, sys
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
M, N, K = int(os.environ.get("ABL_M", "2048")), 16384, 4096
W = torch.randn(N, K, device=dev) * 0.02; W[torch.rand(N, K, device=dev) < 0.005] *= 16
X = torch.randn(M, K, device=dev).to(torch.bfloat16)
res = []
for fo in ("posit8_es1", "fp8_e4m3"):
    P = qlinear.pack_weight(W, 8, 8, "fp4_e2m1", fo, 2, 32, layout="unified")
    res.append(min(t(lambda: qlinear.qlinear(X, P, None, torch.bfloat16)) for _ in range(3)) * 1e3)
print("RESULT %.1f %.1f" % tuple(res))
'''
libs = [("product", None)] + sorted((os.path.basename(p)[len("libmsq_hip_abl"):-3], p) for p in glob.glob(os.path.join(HERE, "abl", "libmsq_hip_abl*.so")))
for mf in os.environ.get("ABL_MFS", "8 16").split():
    for tag, path in libs:
        env = dict(os.environ, MSQ_GEMM_MF=mf)
        if path: env["MSQ_LIB_OVERRIDE"] = path
        out = subprocess.run([sys.executable, "-c", CHILD, os.path.abspath(__file__)], env=env, capture_output=True, text=True, timeout=300)
        line = [l for l in out.stdout.splitlines() if l.startswith("RESULT")]
        print(f"MF={mf} MSQ_ABL={tag:8s}: posit / fp8 GEMM us: {line[0][7:] if line else out.stderr[-300:]}", flush=True)
