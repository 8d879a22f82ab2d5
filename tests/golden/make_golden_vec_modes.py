import os, sys, warnings
os.environ["PYTHONDONTWRITEBYTECODE"] = "1"; sys.dont_write_bytecode = True; warnings.filterwarnings("ignore")
import numpy as np, torch
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE); sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import make_golden as MG
scratch, quant, mx_ops, elemwise_ops, formats, linear, specs, posit_mod = MG._import_reference()
from mx import RMSNorm
from oracle import oracle as O
g = torch.Generator().manual_seed(5)
out = {}
nbad = 0
for bf, rd in ((12, "floor"), (16, "floor"), (12, "nearest")):
    sp = specs.finalize_mx_specs({"w_elem_format": "fp4_e2m1", "a_elem_format": "fp8_e4m3", "scale_bits": 8, "block_size": 32, "bfloat": bf, "round": rd})
    bits = bf - 7; mn = 2.0 ** 127 * (2 ** (bits - 1) - 1) / 2 ** (bits - 2)
    for H in (200, 1024):
        for sc in (2.0 ** -30, 2.0 ** -8, 1.0, 2.0 ** 30):
            for eps in (1e-6, 1e-12):
                x = torch.randn(6, H, generator=g) * sc; w = torch.randn(H, generator=g) * 0.5 + 1; b = torch.randn(H, generator=g) * 0.1
                rn = RMSNorm(H, eps=eps, mx_specs=sp)
                with torch.no_grad():
                    rn.weight.copy_(w); rn.bias.copy_(b); y = rn(x).numpy()
                yo = O.vec_rmsnorm(x.numpy(), w.numpy(), b.numpy(), eps, bits, 8, mn, rd)
                d = int((y != yo).sum())
                nbad += d > 0
                k = "%d|%s|%d|%g|%g" % (bf, rd, H, sc, eps)
                out[k + "|x"], out[k + "|w"], out[k + "|b"], out[k + "|y"] = x.numpy(), w.numpy(), b.numpy(), y
                if d: print("oracle != reference", k, d)
print("cases with oracle != reference:", nbad)
np.savez_compressed(os.path.join(HERE, "vec_rmsnorm_modes.npz"), **out)      # beside itself, like every other generator
