# accumulation accuracy of the scaled MFMA path: |y - exact| against max|y| and against the sum of |products|
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, msq
from msq import qlinear
dev = torch.device("cuda:0")
for seed in range(3):
    g = torch.Generator(device=dev).manual_seed(seed)
    for (N, K, M) in [(256, 512, 300), (4096, 4096, 512), (4096, 11008, 256)]:
        W = torch.randn(N, K, generator=g, device=dev) * 0.02
        W[torch.rand(N, K, generator=g, device=dev) < 0.01] *= 20
        X = torch.randn(M, K, generator=g, device=dev)
        X[torch.rand(M, K, generator=g, device=dev) < 0.02] *= 10
        Xq = msq.mx_ops._quantize_mx(X, 8, "fp8_e4m3", axes=[-1], block_size=32).double()
        for name in ("w4", "w8"):
            if name == "w4":
                P = qlinear.mx_pack_weight(W); Wq = msq.mx_ops._quantize_mx(W, 8, "fp4_e2m1", axes=[-1], block_size=32).double()
            else:
                Wq = msq.quant.outlier_fakequant(W, 8, 8, "fp4_e2m1", "fp8_e4m3", 2, -1, 32)["out"]
                P = qlinear.mx_pack_values(Wq); Wq = Wq.double()
            y = qlinear.qlinear_mx_w4a8(X, P, None, torch.float32).double()
            ref = Xq @ Wq.t(); ab = Xq.abs() @ Wq.abs().t()
            err = (y - ref).abs()
            print(f"seed {seed} N{N} K{K} M{M} {name}: max err {err.max().item():.3e} / max|y| = {err.max().item()/ref.abs().max().item():.2e}; "
                  f"max err / sum|products| = {(err/ab).max().item():.2e} (2^{torch.log2((err/ab).max()).item():.1f})", flush=True)
