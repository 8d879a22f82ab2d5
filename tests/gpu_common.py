"""Helpers shared by the row-named GPU test files (no tests here)."""
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")


def dev():
    return torch.device("cuda:0")


def weights(N, K, seed=0):
    """the BASELINE synthetic weight: randn * 0.02 with 1 % of the entries x16"""
    g = torch.Generator().manual_seed(seed)
    W = torch.randn(N, K, generator=g) * 0.02
    W[torch.rand(N, K, generator=g) < 0.01] *= 16
    return W


def eq(a, b):
    """equal values (a zero of either sign is the same value; NaN equals NaN)"""
    a = np.asarray(a, np.float32); b = np.asarray(b, np.float32)
    return bool(((a == b) | (np.isnan(a) & np.isnan(b))).all())


def planted(rows, K, seed, lo=-12, hi=12):
    """[rows, K] float32 whose blocks of 32 each hold ONE maximum 1 .. 90 ulps under a power of two 2^u (u in lo .. hi, either sign) and
    smaller values elsewhere: floor(torch.log2(max)) is u for the nearest of them, u - 1 beyond -- the block scale of the reference's
    Python path doubles exactly there, the native kernel's exponent field never does."""
    rs = np.random.RandomState(seed)
    x = rs.randn(rows, K).astype(np.float32)
    nb = K // 32
    for r in range(rows):
        for b in range(nb):
            u = int(rs.randint(lo, hi + 1))
            d = int(rs.randint(1, 91)) if rs.rand() < 0.3 else int(rs.randint(1, 7))
            top = np.uint32(np.float32(2.0 ** u).view(np.uint32) - np.uint32(d)).view(np.float32)
            blk = x[r, b * 32:(b + 1) * 32]
            blk *= np.float32(0.45 * 2.0 ** u / max(np.abs(blk).max(), 1e-30))
            blk[rs.randint(0, 32)] = top if rs.rand() < 0.5 else -top
    return x
