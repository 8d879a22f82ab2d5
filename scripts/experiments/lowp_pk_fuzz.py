#!/usr/bin/env python3
"""The packed in-dtype fake-quant kernels (k_outlier_lowp_pk / _pk2, csrc/msq_quant_lowp.hip) against the op-by-op kernel
k_outlier_lowp (MSQ_OUTLIER_LOWP_PK=0: pinned by the reference-made goldens and the oracle) on inputs built to hit what the
packed form argues away: magnitudes one ulp either side of every tie and of the excepted magnitude, scales at the edge of the
exponent bounds, fp16 subnormals, blocks without outliers / without inliers, all-negative blocks, constant blocks (std = 0),
bounds that are exactly zero, NaN / Inf members.  Values, masks and both exponents must agree bit for bit.

    python scripts/experiments/lowp_pk_fuzz.py [rounds]          # needs the GPU
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import msq  # noqa: E402

L = msq._lib.lib()
dev = torch.device("cuda:0")
POSIT = []          # (posit outliers on the packed float32-semantics path were tried and dropped: scripts/experiments/README.md)
COMBOS = [("int2", "fp4"), ("fp4_e2m1", "fp8_e4m3"), ("fp4_e2m1", "fp4_e2m1"), ("fp4_e2m1", "fp8_e5m2"), ("fp8_e4m3", "fp8_e4m3")]


HANDED = {}
DECLINED = {}


def run(W, fi, fo, sd, axis, bs, sb, pk, tag=None, native=True):
    """msq_outlier_fakequant in the tensor's dtype through the C ABI with a workspace of our own: its head is the number of waves the packed
    kernels handed back to the op-by-op kernel (counted per kind of input: a fuzz that never stays on the packed path proves nothing).
    native=False: the tensor computed in FLOAT32 (dtype 1 / 2): the float32-semantics packed kernels against the float32 kernels"""
    from msq._lib import ptr, check, current_stream
    from msq.formats import format_id
    assert L.msq_set_tuning(b"MSQ_OUTLIER_LOWP_PK", pk) == 0
    try:
        x = W.contiguous()
        axis = axis % x.ndim
        pre = int(np.prod(x.shape[:axis])) if axis else 1
        al = x.shape[axis]
        post = int(np.prod(x.shape[axis + 1:])) if axis + 1 < x.ndim else 1
        nblk = (al + bs - 1) // bs
        out = torch.empty_like(x)
        mask = torch.empty(x.shape, dtype=torch.uint8, device=dev)
        e_in = torch.empty((pre, nblk, post), dtype=torch.float32, device=dev)
        e_out = torch.empty((pre, nblk, post), dtype=torch.float32, device=dev)
        st = torch.zeros(1, dtype=torch.int32, device=dev)
        wsb = L.msq_outlier_workspace_bytes(pre, al, post, bs, 0)
        ws = torch.full((wsb // 8 + 1,), -7, dtype=torch.int64, device=dev)          # (the packed kernels clear the head: -7 left = the launcher declined)
        check(L.msq_outlier_fakequant(ptr(x), ptr(out), ptr(mask), ptr(e_in), ptr(e_out), None, ptr(st), ptr(ws), wsb, ((0x11 if x.dtype == torch.float16 else 0x12) if native else (1 if x.dtype == torch.float16 else 2)),
                                      pre, al, post, bs, format_id(fi), format_id(fo), sb, sb, float(sd), 0, 0, 0, current_stream(dev)), "msq_outlier_fakequant")
        if pk and tag is not None:
            h = HANDED.setdefault(tag, [0, 0])
            c = int(ws[0].item())
            nwv = (pre * nblk * post + 63) // 64
            mk = ws.view(torch.uint8)[64 + 8 * (nwv + 2): 64 + 8 * (nwv + 2) + nwv]         # one mark per wave (the list itself stops at 1 / 16 of them)
            h[0] += nwv if c == -7 else int(mk.sum().item()); h[1] += nwv          # declined = every wave on the old kernels
            if c == -7:
                DECLINED[(fi, fo, bs, native)] = DECLINED.get((fi, fo, bs, native), 0) + 1
        return {"out": out, "mask": mask, "e_in": e_in, "e_out": e_out, "status": int(st.item())}
    finally:
        L.msq_set_tuning(b"MSQ_OUTLIER_LOWP_PK", 1)


def same(a, b):
    it = torch.int32 if a.dtype == torch.float32 else torch.int16
    ia, ib = a.view(it), b.view(it)
    nan = torch.isnan(a) & torch.isnan(b)
    return bool(((ia == ib) | nan).all())


def _bump(x, g):
    """two ulps either way on the bit patterns (zeros and the values next to them stay: -1 as a bit pattern is a NaN)"""
    b = torch.randint(-2, 3, x.shape, generator=g, device=dev).to(torch.int16)
    i = x.view(torch.int16)
    keep = (i & 0x7FFF) < 4
    return torch.where(keep, i, i + b).view(x.dtype)


def make(kind, shape, dt, g):
    R, C = shape
    x = torch.randn(R, C, generator=g, device=dev)
    if kind == "weights":
        x = x * 0.02
        x[torch.rand(R, C, generator=g, device=dev) < 0.005] *= 16
    elif kind == "scales":                                       # every row its own binade, through the whole exponent range
        lo, hi = (-24, 15) if dt == torch.float16 else (-60, 60)
        e = torch.randint(lo, hi + 1, (R, 1), generator=g, device=dev).float()
        x = x * torch.exp2(e)
    elif kind == "columns":                                      # the same along the other axis
        lo, hi = (-24, 15) if dt == torch.float16 else (-60, 60)
        e = torch.randint(lo, hi + 1, (1, C), generator=g, device=dev).float()
        x = x * torch.exp2(e)
    elif kind == "ties":                                         # k / 32 2^e +- one ulp of T: every tie and every grid point of the formats
        k = torch.randint(0, 513, (R, C), generator=g, device=dev).float()
        e = torch.randint(-12, 3, (R, 1), generator=g, device=dev).float()
        x = (k / 32.0) * torch.exp2(e) * torch.sign(x)
        x = x.to(dt)
        return _bump(x, g)
    elif kind == "tiesc":
        k = torch.randint(0, 513, (R, C), generator=g, device=dev).float()
        e = torch.randint(-12, 3, (1, C), generator=g, device=dev).float()
        x = (k / 32.0) * torch.exp2(e) * torch.sign(x)
        x = x.to(dt)
        return _bump(x, g)
    elif kind == "negative":
        x = -x.abs() * 0.02
    elif kind == "positive":
        x = x.abs() * 0.3
    elif kind == "sparse":
        x = x * (torch.rand(R, C, generator=g, device=dev) < 0.1)
    elif kind == "constant":
        x = torch.full((R, C), 0.37, device=dev) * torch.sign(x)
        x[::3] = 0.0
    elif kind == "subnormal":
        x = x * 3e-6
    elif kind == "huge":
        x = x * (2.0e4 if dt == torch.float16 else 1e30)
    elif kind == "special":
        x = x * 0.05
        x[5, 7] = float("nan"); x[70, 3] = float("inf"); x[R - 13, C - 9] = -float("inf")
    return x.to(dt)


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    g = torch.Generator(device=dev).manual_seed(1234)
    kinds = ["weights", "scales", "columns", "ties", "tiesc", "negative", "positive", "sparse", "constant", "subnormal", "huge", "special"]
    bad = n = 0
    t0 = time.time()
    for r in range(rounds):
        for dt in (torch.float16, torch.bfloat16):
            for kind in kinds:
                for shape in ((256, 512), (96, 160), (2048, 1024)):
                    if shape[0] == 2048 and r > 0:
                        continue
                    W = make(kind, shape, dt, g)
                    for fi, fo in COMBOS + POSIT:
                        for axis, bs in ((0, 16), (-1, 32), (0, 32), (-1, 16), (0, 8), (-1, 8), (0, 64), (-1, 64)):
                            if W.shape[axis] % bs:
                                continue
                            for sd, sb, native in ((2.0, 8, True), (3.0, 8, True), (1.0, 4, True), (2.0, 8, False), (3.0, 8, False), (1.0, 4, False)):
                                if (not native and bs == 64) or (native and (fi, fo) in POSIT):
                                    continue                 # (in-dtype arithmetic has no posit codec: the reference's harness upcasts there too)
                                a = run(W, fi, fo, sd, axis, bs, sb, 1, (str(dt)[6:] + ("" if native else " (float32 semantics)"), kind), native)
                                b = run(W, fi, fo, sd, axis, bs, sb, 0, None, native)
                                ok = same(a["out"], b["out"]) and torch.equal(a["mask"], b["mask"]) and same(a["e_in"], b["e_in"]) and same(a["e_out"], b["e_out"]) and a["status"] == b["status"]
                                n += 1
                                if not ok:
                                    bad += 1
                                    dv = int(((a["out"].view(torch.int16) != b["out"].view(torch.int16)) & ~(torch.isnan(a["out"]) & torch.isnan(b["out"]))).sum())
                                    dm = int((a["mask"] != b["mask"]).sum())
                                    print("MISMATCH", "native" if native else "float32-semantics", str(dt)[6:], kind, shape, fi, fo, "axis", axis, "bs", bs, "k", sd, "sb", sb, "values", dv, "masks", dm, flush=True)
                                    if bad <= 6:
                                        idx = ((a["out"].view(torch.int16) != b["out"].view(torch.int16)) & ~(torch.isnan(a["out"]) & torch.isnan(b["out"]))).nonzero()[:4]
                                        for i in idx.tolist():
                                            print("   at", i, "x", float(W[i[0], i[1]]), hex(int(W.view(torch.int16)[i[0], i[1]]) & 0xFFFF), "packed", float(a["out"][i[0], i[1]]),
                                                  "op-by-op", float(b["out"][i[0], i[1]]), "mask", int(a["mask"][i[0], i[1]]), int(b["mask"][i[0], i[1]]))
    for (dn, kind), (h, w) in sorted(HANDED.items()):
        print("%-30s %-10s waves handed back to the op-by-op kernel: %8d of %9d (%.2f %%)" % (dn, kind, h, w, 100.0 * h / max(w, 1)))
    for k, v in sorted(DECLINED.items(), key=str):
        print("launcher declined (ragged axis / odd stride / not a packed format pair):", k, v, "calls")
    print("cases %d, mismatching %d, %.0f s" % (n, bad, time.time() - t0))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
