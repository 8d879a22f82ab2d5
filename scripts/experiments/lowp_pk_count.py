#!/usr/bin/env python3
"""How many waves the packed in-dtype fake-quant kernels (k_outlier_lowp_pk / _pk2) hand back to the op-by-op kernel on a weight-like tensor:
the head of the workspace after the call is the length of their list."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import msq
from msq._lib import lib, ptr, check, current_stream
from msq.formats import format_id
dev = torch.device("cuda:0")
L = lib()
torch.manual_seed(0)
W = torch.randn(16384, 4096, device=dev) * 0.02
W[torch.rand(16384, 4096, device=dev) < 0.005] *= 16
for dt, code in ((torch.float16, 0x11), (torch.bfloat16, 0x12)):
    x = W.to(dt)
    for fi, fo, axis, bs in (("int2", "fp4", 0, 16), ("fp4_e2m1", "fp8_e4m3", 1, 32), ("fp4_e2m1", "fp8_e4m3", 0, 32), ("int2", "fp4", 1, 16)):
        pre, al, post = (1, 16384, 4096) if axis == 0 else (16384, 4096, 1)
        out = torch.empty_like(x)
        st = torch.zeros(1, dtype=torch.int32, device=dev)
        wsb = L.msq_outlier_workspace_bytes(pre, al, post, bs, 0)
        ws = torch.empty(wsb // 8 + 1, dtype=torch.int64, device=dev)
        check(L.msq_outlier_fakequant(ptr(x), ptr(out), None, None, None, None, ptr(st), ptr(ws), wsb, code, pre, al, post, bs, format_id(fi), format_id(fo), 8, 8, 2.0, 0, 0, 0,
                                      current_stream(dev)), "fq")
        v = int(ws[0].item())                       # (the list; it stops at 1 / 16 of the waves -- the marks behind it are complete)
        waves = (pre * (al // bs) * post + 63) // 64
        print(str(dt)[6:], fi, fo, "axis", axis, "bs", bs, ": handed back %d of %d waves (%.3f %%), status bits %d" % (v, waves, 100.0 * v / waves, int(st.item())))
