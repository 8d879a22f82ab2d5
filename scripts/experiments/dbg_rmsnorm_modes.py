"""Debugging aid (round 5): RMSNorm under bfloat12 / floor rounding on tiny inputs against the reference-made fixture, stage by stage."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, msq
from msq import vector_ops as V
from oracle import oracle as O
dev = torch.device("cuda:0")
z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests", "golden", "vec_rmsnorm_modes.npz"))
k = "12|floor|200|9.31323e-10|1e-06"
specs = msq.specs.finalize_mx_specs({"w_elem_format": "fp4_e2m1", "a_elem_format": "fp8_e4m3", "scale_bits": 8, "block_size": 32, "custom_cuda": True, "bfloat": 12, "round": "floor"})
mn = 2.0 ** 127 * (2 ** 4 - 1) / 2 ** 3
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
x, w, b, yref = z[k + "|x"], z[k + "|w"], z[k + "|b"], z[k + "|y"]
one, zero = np.ones_like(w), np.zeros_like(b)
for name, ww, bb in (("w=1,b=0", one, zero), ("w,b=0", w, zero), ("w=1,b", one, b), ("w,b", w, b)):
    y = V.rms_norm(t(x), t(ww), t(bb), 1e-6, specs).cpu().numpy()
    yo = O.vec_rmsnorm(x, ww, bb, 1e-6, 5, 8, mn, "floor")
    d = np.argwhere(y != yo)
    print(name, "differ", len(d))
    for i in d[:3]:
        i = tuple(i)
        print("   x %r w %r b %r -> gpu %r oracle %r" % (float(x[i]), float(ww[i[1]]), float(bb[i[1]]), float(y[i]), float(yo[i])))
# the rounding itself on the values involved
v = np.concatenate([x.ravel(), (x * 1024).ravel(), b.ravel(), w.ravel()]).astype(np.float32)
out = torch.empty(v.size, device=dev)
from msq._lib import lib, ptr, current_stream
tv = t(v)
lib().msq_vec_round(ptr(tv), ptr(out), v.size, 5, 8, mn, 1, 1, 0, current_stream(dev))
ro = O.vec_add(v, np.zeros_like(v), 5, 8, mn, "floor") if False else None
import ctypes
print("round floor bf12: gpu vs codec-forced:", end=" ")
out2 = torch.empty(v.size, device=dev)
lib().msq_vec_round(ptr(tv), ptr(out2), v.size, 5, 8, mn, 1, 1, 1, current_stream(dev))
print(int((out != out2).sum().item()), "differences")
# simd_add on the same operands: Q(Q(xs) + Q(b))
xs = (x * 1024).astype(np.float32)
bb2 = np.broadcast_to(b, x.shape).astype(np.float32).copy()
ya = V.simd_add(t(xs), t(bb2), mx_specs=specs).cpu().numpy()
yo = O.vec_add(xs, bb2, 5, 8, mn, "floor")
d = np.argwhere(ya != yo); print("simd_add differ", len(d))
for i in d[:3]:
    i = tuple(i); print("   ", float(xs[i]), float(bb2[i]), "gpu", float(ya[i]), "oracle", float(yo[i]))
# plain torch float32 add on the GPU and on the CPU of the rounded operands
qa = torch.empty(xs.size, device=dev); qb = torch.empty(xs.size, device=dev)
lib().msq_vec_round(ptr(t(xs.ravel())), ptr(qa), xs.size, 5, 8, mn, 1, 1, 0, current_stream(dev))
lib().msq_vec_round(ptr(t(bb2.ravel())), ptr(qb), xs.size, 5, 8, mn, 1, 1, 0, current_stream(dev))
sg = (qa + qb).cpu().numpy(); sc_ = qa.cpu().numpy() + qb.cpu().numpy()
print("float32 sums GPU vs CPU differ:", int((sg != sc_).sum()))
