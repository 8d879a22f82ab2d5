import sys, os
sys.path.insert(0, '/root/repo')
import torch, msq
from msq import qlinear
dev = torch.device("cuda:0"); torch.manual_seed(0)
M,K=2048,4096
X = torch.randn(M,K,device=dev)
def t(fn,n=30):
    fn(); torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n
for variant,sd in ((0,2),(1,5)):
    ms=t(lambda: qlinear.act_quant(X,8,8,"fp8_e4m3","fp8_e4m3",sd,32,"nearest",False,variant))
    print(f"act_quant variant {variant}: {ms*1e3:.1f} us  {M*K*6/ms/1e6:.0f} GB/s")
