# decode latency (M <= 16), HIP events, 5 warm-up calls
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, msq
from msq import qlinear
dev = torch.device("cuda:0"); torch.manual_seed(0)
def t(fn,n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n
for (N,K) in [(16384,4096),(4096,4096),(11008,4096),(4096,11008)]:
    W = torch.randn(N,K,device=dev)*0.02
    for fo in ("fp8_e4m3","posit8_es1"):
        P = qlinear.pack_weight(W,8,8,"fp4_e2m1",fo,2,32,layout="unified")
        Wu = qlinear.unpack_weight(P, torch.bfloat16)
        for M in (1,8,16):
            X = torch.randn(M,K,device=dev).to(torch.bfloat16)
            ms = min(t(lambda: qlinear.qlinear(X,P)) for _ in range(3))
            mb = min(t(lambda: X @ Wu.t()) for _ in range(3))
            print(f"N{N} K{K} {fo} M{M}: {ms*1e3:.1f} us ({P.nbytes/ms/1e6:.0f} GB/s packed) | hipBLASLt bf16 {mb*1e3:.1f} us", flush=True)

# the same inside a HIP graph (what a decode loop would replay): no host launch overhead
print("# HIP-graph replay (20 decode launches per graph)")
for (N,K) in [(16384,4096),(4096,4096),(11008,4096),(4096,11008)]:
    W = torch.randn(N,K,device=dev)*0.02
    P = qlinear.pack_weight(W,8,8,"fp4_e2m1","fp8_e4m3",2,32,layout="unified")
    Wu = qlinear.unpack_weight(P, torch.bfloat16)
    for M in (1,16):
        X = torch.randn(M,K,device=dev).to(torch.bfloat16)
        res = {}
        for name, fn in (("fused", lambda: qlinear.qlinear(X,P)), ("hipBLASLt bf16", lambda: X @ Wu.t())):
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                for _ in range(3): fn()
                s.synchronize()
                gph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gph, stream=s):
                    for _ in range(20): y = fn()
            torch.cuda.synchronize()
            def run():
                gph.replay()
            res[name] = min(t(run, 20) for _ in range(3)) / 20
        print(f"N{N} K{K} M{M}: fused {res['fused']*1e3:.1f} us ({P.nbytes/res['fused']/1e6:.0f} GB/s packed) | hipBLASLt bf16 {res['hipBLASLt bf16']*1e3:.1f} us", flush=True)
