"""MX activation packer (msq_mx_pack_a8*: fp32 / bf16 / fp16 in, e4m3 codes + E8M0 scale bytes out) alone: device time from graph replays,
bytes moved and the HBM fraction, at the W4A8 bench shape and two more."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import msq, bench
from msq import qlinear
dev = torch.device("cuda:0")
for M, K in ((2048, 4096), (2048, 11008), (8192, 4096)):
    for dt in (torch.float32, torch.bfloat16, torch.float16):
        X = torch.randn(M, K, device=dev).to(dt)
        for _ in range(10): qlinear.mx_pack_act(X)
        ms = bench._tgraph([lambda: qlinear.mx_pack_act(X)] * 10)
        byts = X.numel() * X.element_size() + X.numel() + X.numel() // 32
        print("M%5d K%6d %-9s %6.1f us  %6.1f MB  %5.2f TB/s  (%.2f of 8 TB/s)" % (M, K, str(dt)[6:], ms * 1e3, byts / 1e6, byts / ms / 1e9, byts / ms / 1e9 / 8), flush=True)
