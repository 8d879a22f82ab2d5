"""Every GEMM form on given shapes (MSQ_GEMM_256 through msq_set_tuning: rules / 0 = k_qgemm3 / 1 = 256-row / 2 = 128-row / 3 = persistent), device time from
HIP-graph replays.  SHAPES="M,N,K;..." """
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import msq
from msq import qlinear
from msq._lib import lib
dev = torch.device("cuda:0")
INT_MIN = -2 ** 31
shapes = [tuple(int(v) for v in t.split(",")) for t in os.environ.get("SHAPES", "1024,8192,28672;2048,8192,28672;1024,8192,3584;2048,8192,3584").split(";")]
def tg(fn, n=10, reps=5):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    out = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / n * 1e3)
    return sorted(out)[len(out) // 2]
for (M, N, K) in shapes:
    W = torch.randn(N, K, device=dev) * 0.02
    P = qlinear.pack_weight(W, 8, 8, "fp4_e2m1", "posit8_es1", 2, 32, layout="unified")
    del W
    X = torch.randn(M, K, device=dev).to(torch.bfloat16)
    for ydt in (torch.bfloat16, torch.float32):
        row = []
        for form in (INT_MIN, 0, 1, 2, 3):
            lib().msq_set_tuning(b"MSQ_GEMM_256", form)
            try:
                us = tg(lambda: qlinear.qlinear(X, P, None, ydt))
                row.append("%s %.1f us (%.3f)" % ({INT_MIN: "rules", 0: "k_qgemm3", 1: "256-row", 2: "128-row", 3: "persistent"}[form], us, 2.0 * M * N * K / us / 1e6 / 2500))
            except Exception as e:
                row.append("form %d: %s" % (form, str(e)[:40]))
        lib().msq_set_tuning(b"MSQ_GEMM_256", INT_MIN)
        print("M%d N%d K%d %s: " % (M, N, K, str(ydt)[6:]) + " | ".join(row), flush=True)
