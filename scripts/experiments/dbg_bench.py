"""debug aid: which step of bench.layer7b_prefill stalls (faulthandler dump every 40 s)"""
import faulthandler, sys, time, os
faulthandler.dump_traceback_later(40, repeat=True, file=sys.stderr)
sys.path.insert(0, os.getcwd())
import torch, bench
from msq import qlinear
dev = torch.device("cuda:0")
t0 = time.time()
def say(*a):
    torch.cuda.synchronize(); print("%.1f" % (time.time() - t0), *a, flush=True)
M = 2048
X = {k_: torch.randn(M, k_, device=dev).to(torch.bfloat16) for k_ in (4096, 11008)}
for name, n_, k_ in (("qkv", 12288, 4096), ("o", 4096, 4096)):
    W = bench.synth_weight(n_, k_, dev, seed=3); say(name, "weight")
    for fo in ("posit8_es1", "fp8_e4m3"):
        P = qlinear.pack_weight(W, 8, 8, "fp4_e2m1", fo, 2, 32, layout="unified"); say(fo, "packed", bench._kernel_name(P, M, False))
        for _ in range(20):
            qlinear.qlinear(X[k_], P, None, torch.bfloat16)
        say("warm")
        ms = bench._tgraph([lambda P=P: qlinear.qlinear(X[k_], P, None, torch.bfloat16)] * 10); say("graph", ms * 1e3)
    dense = qlinear.unpack_weight(P, torch.bfloat16); say("unpacked")
    del P
    y = X[k_] @ dense.t(); say("one matmul")
    for _ in range(20):
        X[k_] @ dense.t()
    say("20 matmuls")
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        X[k_] @ dense.t()
        st.synchronize()
    say("side stream matmul")
    ms = bench._tgraph([lambda: X[k_] @ dense.t()] * 10); say("graph matmul", ms * 1e3)
