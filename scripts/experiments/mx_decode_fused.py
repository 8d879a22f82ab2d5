"""Round 3 experiment (NOT in the product): MX decode (M = 1 ... 32) from a HIP graph with the activation quantiser fused into the
decode kernel (every wave quantised its own 128-k chunk of the raw activations in registers: 32 exponent extractions, two shuffles,
16 scaled converts per tile and row group; an experimental `msq_qlinear_mx_xq` entry point, not kept) against activation pack +
decode kernel.  Result on MI355X, MX-FP4, N16384 K4096: M = 1 fused 9.1 (fp32 input) / 10.8 us (bf16) against 9.6-9.9 us for pack +
kernel (kernel alone 7.1); M = 16: 11.6-13.6 against 11.7; M = 32: 49-52 against 25.6.  The quantisation is repeated by each of the
256 column strips (x 16 rows per fragment), which costs as much as the pack launch it saves at M = 1 and far more beyond: dropped.
What would pay is a pack fused into the PRODUCER of the activations (the norm / residual add in front of the projections), which is
outside this hot path.  (The script needs that experimental build to run the "fused" arm.)"""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import msq
from msq import qlinear
dev = torch.device("cuda:0"); torch.manual_seed(0)
def graph_time(fn, n=50):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(10): fn()
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n / 10 * 1e3
for (N, K) in ((16384, 4096), (4096, 4096), (11008, 4096), (4096, 11008), (12288, 4096), (22016, 4096)):
    W = torch.randn(N, K, device=dev) * 0.02
    for tag, P in (("mx-fp4", qlinear.mx_pack_weight(W)),):
        for M in (1, 16, 32):
            for dt in (torch.bfloat16, torch.float32):
                x = torch.randn(M, K, device=dev).to(dt)
                t_f = graph_time(lambda: qlinear.qlinear_mx_w4a8(x, P, None, torch.bfloat16))
                t_2 = graph_time(lambda: qlinear.qlinear_mx_w4a8(qlinear.mx_pack_act(x), P, None, torch.bfloat16))
                xp = qlinear.mx_pack_act(x)
                t_k = graph_time(lambda: qlinear.qlinear_mx_w4a8(xp, P, None, torch.bfloat16))
                print("N%6d K%6d M%3d %-8s %-8s: fused %.1f us | pack + kernel %.1f us | kernel alone on packed activations %.1f us" % (N, K, M, tag, str(dt)[6:], t_f, t_2, t_k), flush=True)
