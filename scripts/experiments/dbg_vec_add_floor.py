import sys; sys.path.insert(0, "/root/repo")
import numpy as np, torch, msq, struct
from msq._lib import lib, ptr, current_stream
from oracle import oracle as O
dev = torch.device("cuda:0")
mn = 2.0 ** 127 * (2 ** 4 - 1) / 2 ** 3
a = np.array([-5.344744646862409e-10] * 8, np.float32); b = np.array([0.01577991619706154] * 8, np.float32)
def h(v): return hex(struct.unpack("<I", struct.pack("<f", float(v)))[0])
for n in (1, 4, 8):
    ta, tb = torch.from_numpy(a[:n]).to(dev), torch.from_numpy(b[:n]).to(dev)
    out = torch.empty(n, device=dev)
    for rm in (0, 1, 2):
        lib().msq_vec_add(ptr(ta), ptr(tb), 0.0, ptr(out), n, 5, 8, mn, rm, 1, current_stream(dev))
        o = O.vec_add(a[:n], b[:n], 5, 8, mn, ["nearest", "floor", "even"][rm])
        print("n", n, "rmode", rm, "gpu", h(out[0].item()), "oracle", h(o[0]))
qa = torch.empty(1, device=dev); lib().msq_vec_round(ptr(torch.from_numpy(a[:1]).to(dev)), ptr(qa), 1, 5, 8, mn, 1, 1, 0, current_stream(dev)); print("Q(a) floor", h(qa.item()))
qb = torch.empty(1, device=dev); lib().msq_vec_round(ptr(torch.from_numpy(b[:1]).to(dev)), ptr(qb), 1, 5, 8, mn, 1, 1, 0, current_stream(dev)); print("Q(b) floor", h(qb.item()))
s = (qa + qb); print("sum", h(s.item()))
qs = torch.empty(1, device=dev); lib().msq_vec_round(ptr(s), ptr(qs), 1, 5, 8, mn, 1, 1, 0, current_stream(dev)); print("Q(sum) floor", h(qs.item()))
