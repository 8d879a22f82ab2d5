#!/usr/bin/env python3
"""half_away_quirk.npz: the reference's own outputs on planted pred(half the smallest step) values (see tests/test_gpu_a2_half_away_quirk.py).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_quirk.py

Imports the reference from a scratch copy of /root/reference (build container only); stores inputs and the reference's outputs, nothing else.
  mx|<fmt>|<round>     : mx_ops._quantize_mx(A, 8, fmt, axes=[-1], block_size=32, round)            (Python path, `+1e-6` divisor included)
  a6|<fi>|<fo>|ax<a>|bs<b> : utils.quant.quantize_mx_outlier_v1(A, 8, 8, fi, fo, 'max', 2, [a], b, 'nearest')"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import _import_reference  # noqa: E402


def pred(x, n=1):
    return np.uint32(np.float32(x).view(np.uint32) - np.uint32(n)).view(np.float32)


def succ(x):
    return np.uint32(np.float32(x).view(np.uint32) + np.uint32(1)).view(np.float32)


def plant_rows(K, block, half_step_exp_of, seed, urange=(-6, 7)):
    rs = np.random.RandomState(seed)
    rows = []
    for u in range(*urange):
        r = (rs.randn(K) * 1e-3).astype(np.float32)
        for b in range(K // block):
            top = np.float32((1.0 + 0.75 * rs.rand()) * 2.0 ** (u + b % 3))
            h = np.float32(2.0 ** half_step_exp_of(int(np.floor(np.log2(top)))))
            blk = r[b * block:(b + 1) * block]
            blk *= np.float32(h * 4)
            blk[0] = top if b % 2 else -top
            blk[1:6] = [pred(h), -pred(h), h, pred(h, 2), succ(h)]
        rows.append(r)
    return np.stack(rows)


def main():
    scratch, quant, mx_ops, elemwise_ops, formats, linear, specs, posit_mod = _import_reference()
    out = {}
    for fmt, emax, half in (("fp4_e2m1", 2, -2), ("fp8_e4m3", 8, -10), ("fp6_e3m2", 4, -5), ("int4", 0, -3)):
        A = plant_rows(128, 32, lambda e: e - emax + half, 3)
        out["in|mx|%s" % fmt] = A
        for rnd in ("nearest", "even", "floor"):
            y = mx_ops._quantize_mx(torch.from_numpy(A.copy()), 8, fmt, axes=[-1], block_size=32, round=rnd)
            out["mx|%s|%s" % (fmt, rnd)] = y.numpy()
    for fi, fo, emax, half in (("fp4_e2m1", "fp8_e4m3", 2, -2), ("int2", "fp4", 0, -1)):
        for axis, bs in ((-1, 32), (0, 16)):
            rs = np.random.RandomState(9)
            A = plant_rows(128, bs, lambda e: e - emax + half, 5)
            flat = ((1.0 + 0.2 * rs.rand(*A.shape)) * 0.75).astype(np.float32)
            h = np.float32(2.0 ** (-1 - emax + half))
            flat[:, 1::bs] = pred(h); flat[:, 2::bs] = -pred(h); flat[:, 3::bs] = h
            A = np.concatenate([A, flat])
            if axis == 0:
                A = np.ascontiguousarray(A.reshape(A.shape[0], -1, bs).transpose(2, 0, 1).reshape(bs, -1))
            key = "%s|%s|ax%d|bs%d" % (fi, fo, axis, bs)
            out["in|a6|" + key] = A
            y = quant.quantize_mx_outlier_v1(torch.from_numpy(A.copy()), 8, 8, fi, fo, "max", 2, [axis], bs, "nearest")
            out["a6|" + key] = y.numpy()
    np.savez_compressed(os.path.join(HERE, "half_away_quirk.npz"), **out)
    print("wrote half_away_quirk.npz:", len(out), "arrays")


if __name__ == "__main__":
    main()
