import sys, os, ctypes as C, importlib, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
pkg = importlib.import_module("microscopiq-llm-quantization_amd._lib")
L = pkg.lib()
dev = torch.device("cuda:0")
torch.manual_seed(0)
def pack(W, fi, fo, bs):
    N,K = W.shape
    ik, ok = C.c_int(), C.c_int()
    pkg.check(L.msq_packed_kinds(pkg.format_id(fi), pkg.format_id(fo), C.byref(ik), C.byref(ok)))
    ib,ob,sb,wb = C.c_int64(),C.c_int64(),C.c_int64(),C.c_int64()
    pkg.check(L.msq_packed_sizes(N,K,bs,ik.value,ok.value,C.byref(ib),C.byref(ob),C.byref(sb),C.byref(wb)))
    inl = torch.empty(max(ib.value,16), dtype=torch.uint8, device=dev); out = torch.empty(ob.value, dtype=torch.uint8, device=dev)
    scl = torch.empty(max(sb.value,16), dtype=torch.uint8, device=dev); ws = torch.empty(wb.value, dtype=torch.uint8, device=dev)
    st = torch.zeros(1, dtype=torch.int32, device=dev)
    pkg.check(L.msq_outlier_pack(pkg.ptr(W), pkg.ptr(inl), pkg.ptr(out), pkg.ptr(scl), pkg.ptr(st), pkg.ptr(ws), wb.value, N, K, bs,
              pkg.format_id(fi), pkg.format_id(fo), 8, 8, 2.0, 0, 0, pkg.current_stream()), "pack")
    torch.cuda.synchronize()
    return dict(inl=inl,out=out,scl=scl,ik=ik.value,ok=ok.value,bs=bs,N=N,K=K,status=int(st.item()), bytes=ib.value+ob.value+sb.value)
def unpack(P, dtype=torch.float32):
    Wd = torch.empty(P['N'],P['K'],dtype=dtype,device=dev)
    pkg.check(L.msq_outlier_unpack(pkg.ptr(P['inl']),pkg.ptr(P['out']),pkg.ptr(P['scl']),pkg.ptr(Wd), 0 if dtype==torch.float32 else 2, P['N'],P['K'],P['bs'],P['ik'],P['ok'],pkg.current_stream()),"unpack")
    return Wd
def fakequant(W, fi, fo, bs):
    N,K=W.shape; out=torch.empty_like(W)
    pkg.check(L.msq_outlier_fakequant(pkg.ptr(W),pkg.ptr(out),None,None,None,None,None,None,0,0,N,K,1,bs,pkg.format_id(fi),pkg.format_id(fo),8,8,2.0,0,0,0,pkg.current_stream()))
    return out
def qlinear(X, P, bias=None, ydt=torch.bfloat16):
    M,K = X.shape; Y = torch.empty(M,P['N'],dtype=ydt,device=dev)
    pkg.check(L.msq_qlinear_bf16(pkg.ptr(X),pkg.ptr(P['inl']),pkg.ptr(P['out']),pkg.ptr(P['scl']),pkg.ptr(bias),pkg.ptr(Y), 2 if ydt==torch.bfloat16 else 0, M,P['N'],K,P['bs'],P['ik'],P['ok'],pkg.current_stream()),"qlinear")
    return Y
allok=True
for (N,K,bs,fi,fo) in [(256,256,32,"fp4_e2m1","fp8_e4m3"),(512,1024,32,"fp4_e2m1","posit8_es1"),(256,512,16,"int2","fp4"),(256,256,32,"fp4","fp8_e5m2"),(256,512,64,"fp6_e3m2","fp8_e4m3"),(256,256,8,"fp4","fp8_e4m3"),(256,256,128,"fp4","int8")]:
    W = torch.randn(N,K,device=dev)*0.02; W[torch.rand(N,K,device=dev)<0.01]*=20
    P = pack(W,fi,fo,bs); Wq = fakequant(W,fi,fo,bs); Wu = unpack(P); Wb = unpack(P, torch.bfloat16)
    same = bool((Wu==Wq).all()); sameb = bool((Wb.float()==Wq).all())
    M=300
    X = torch.randn(M,K,device=dev).to(torch.bfloat16)
    bias = torch.randn(N,device=dev)
    Y = qlinear(X,P,bias,torch.float32)
    ref = X.float().double() @ Wq.double().t() + bias.double()
    err = (Y.double()-ref).abs().max().item(); scale = ref.abs().max().item()
    Yb = qlinear(X,P,None,torch.bfloat16)
    errb = (Yb.double()-(ref-bias.double())).abs().max().item()
    ok = same and sameb and err < 1e-4*max(scale,1) and P['status']==0
    allok &= ok
    print(N,K,bs,fi,fo,"kinds",P['ik'],P['ok'],"bits/elem %.2f"%(P['bytes']*8/(N*K)),"unpack==fakequant",same,sameb,"status",P['status'],"gemm f32 maxerr %.3e (scale %.2f) bf16out err %.3e"%(err,scale,errb), "OK" if ok else "FAIL")
print("ALL OK" if allok else "SOME FAILED")
# timing at the north-star shape
for (fi,fo) in [("fp4_e2m1","fp8_e4m3"),("fp4_e2m1","posit8_es1"),("int8","int8")]:
    N,K,M = 16384,4096,2048
    W = torch.randn(N,K,device=dev)*0.02; W[torch.rand(N,K,device=dev)<0.005]*=16
    P = pack(W,fi,fo,32)
    X = torch.randn(M,K,device=dev).to(torch.bfloat16)
    Y = qlinear(X,P); torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): Y = qlinear(X,P)
    e1.record(); torch.cuda.synchronize()
    ms=e0.elapsed_time(e1)/20
    print(fi,fo,"M=%d N=%d K=%d: %.3f ms  %.1f TFLOP/s"%(M,N,K,ms,2*M*N*K/ms/1e9), "status",P['status'])
    Wq = unpack(P, torch.bfloat16)
    Yr = X @ Wq.t(); torch.cuda.synchronize()
    e0.record()
    for _ in range(20): Yr = X @ Wq.t()
    e1.record(); torch.cuda.synchronize()
    ms2=e0.elapsed_time(e1)/20
    print("   torch bf16 matmul (hipBLASLt) on dequantised W: %.3f ms %.1f TFLOP/s; max|diff| %.3e"%(ms2,2*M*N*K/ms2/1e9,(Y.float()-Yr.float()).abs().max().item()))
