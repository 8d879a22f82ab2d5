import sys; sys.path.insert(0, "/root/repo")
import numpy as np, torch, msq
from msq import vector_ops as V
from oracle import oracle as O
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(3)
for bf, rd in ((12, "floor"), (16, "floor"), (12, "even"), (12, "nearest")):
    specs = msq.specs.finalize_mx_specs({"w_elem_format": "fp4_e2m1", "a_elem_format": "fp8_e4m3", "scale_bits": 8, "block_size": 32, "custom_cuda": True, "bfloat": bf, "round": rd})
    bits = bf - 7; mn = 2.0 ** 127 * (2 ** (bits - 1) - 1) / 2 ** (bits - 2)
    for H in (200, 1024):
        x = torch.randn(5, H, generator=g); w = torch.randn(H, generator=g) * 0.5 + 1; b = torch.randn(H, generator=g) * 0.1
        y = V.rms_norm(x.to(dev), w.to(dev), b.to(dev), 1e-6, specs).cpu().numpy()
        yo = O.vec_rmsnorm(x.numpy(), w.numpy(), b.numpy(), 1e-6, bits, 8, mn, rd)
        ln = V.layer_norm(x.to(dev), w.to(dev), b.to(dev), 1e-6, specs).cpu().numpy()
        lo = O.vec_layernorm(x.numpy(), w.numpy(), b.numpy(), 1e-6, bits, 8, mn, rd)
        a = torch.randn(7, 64, generator=g) * 3; c = torch.randn(7, 64, generator=g)
        m = V.simd_mul(a.to(dev), c.to(dev), mx_specs=specs).cpu().numpy(); mo = O.vec_mul(a.numpy(), c.numpy(), bits, 8, mn, rd)
        ad = V.simd_add(a.to(dev), c.to(dev), mx_specs=specs).cpu().numpy(); ao = O.vec_add(a.numpy(), c.numpy(), bits, 8, mn, rd)
        s = V.silu(a.to(dev), mx_specs=specs).cpu().numpy(); so = O.vec_silu(a.numpy(), bits, 8, mn, rd)
        print(bf, rd, H, "rms diff", int((y != yo).sum()), "ln diff", int((ln != lo).sum()), "mul", int((m != mo).sum()), "add", int((ad != ao).sum()), "silu", int((s != so).sum()))
        if (m != mo).any():
            i = np.argwhere(m != mo)[0]; print("   mul", a.numpy()[tuple(i)], c.numpy()[tuple(i)], m[tuple(i)], mo[tuple(i)])
        if (y != yo).any():
            i = np.argwhere(y != yo)[0]; print("   rms", y[tuple(i)], yo[tuple(i)])
