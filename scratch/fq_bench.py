import sys, os
sys.path.insert(0, '/root/repo')
import numpy as np, torch, msq
from msq import _lib as pkg
L = pkg.lib(); dev = torch.device("cuda:0")
for (shape, axis, bs, fi, fo) in [((16384,4096),-1,32,"fp4_e2m1","fp8_e4m3"),((16384,4096),0,16,"int2","fp4"),((16384,4096),-1,32,"fp4_e2m1","posit8_es1"),((16384,4096),0,32,"fp4_e2m1","fp8_e4m3"),((16384,4096),-1,16,"int2","fp4")]:
    A = (torch.randn(*shape)*0.02).to(dev); out = torch.empty_like(A)
    ax = axis % 2; pre = int(np.prod(shape[:ax])); post=int(np.prod(shape[ax+1:])); al=shape[ax]
    def call():
        pkg.check(L.msq_outlier_fakequant(pkg.ptr(A), pkg.ptr(out), None,None,None,None,None,None,0,0,pre,al,post,bs,pkg.format_id(fi),pkg.format_id(fo),8,8,2.0,0,0,0,pkg.current_stream()))
    call(); torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): call()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)/20
    print(shape, "axis",axis,"bs", bs, fi, fo, f"{ms*1e3:.1f} us  {2*A.numel()*4/ms/1e6:.0f} GB/s")
