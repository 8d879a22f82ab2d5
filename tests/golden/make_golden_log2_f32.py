#!/usr/bin/env python3
"""floor(torch.log2(x)) on float32 tensors -- what the reference's Python path computes for shared exponents (utils/quant.py:525-529)
and private exponents (number_system/mx/elemwise_ops.py:139-140) -- pinned with torch itself (2.10 CPU, the reference's tensor library;
nothing is imported from /root/reference: the expression is `torch.floor(torch.log2(t))`).

For every float32 binade, subnormal ones included: the 128 largest values below the power of two above it (where torch.log2's rounding
to float32 lifts the floor by one), the 16 smallest values of the binade and 64 random ones; vector path (whole tensor) and scalar
path (one element at a time, spot-checked) agree.  Stored: the bit patterns (uint32) and the floors (int16).

    python tests/golden/make_golden_log2_f32.py
"""
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    rng = np.random.RandomState(0)
    bits = []
    for E in range(1, 255):                                   # normal binades
        top = ((E + 1) << 23) - 1 - np.arange(128, dtype=np.int64)
        low = (E << 23) + np.arange(16, dtype=np.int64)
        rnd = (E << 23) + rng.randint(0, 1 << 23, size=64).astype(np.int64)
        bits += [top, low, rnd]
    for p in range(0, 23):                                    # subnormal binades: leading one at bit p
        lo, hi = 1 << p, (2 << p) - 1
        top = hi - np.arange(min(128, hi - lo + 1), dtype=np.int64)
        low = lo + np.arange(min(16, hi - lo + 1), dtype=np.int64)
        bits += [top, low]
    u = np.unique(np.concatenate(bits)).astype(np.uint32)
    x = torch.from_numpy(u.view(np.float32).copy())
    torch.set_num_threads(1)
    fl = torch.floor(torch.log2(x)).numpy()
    idx = rng.choice(len(u), size=4000, replace=False)        # scalar path: one element per call
    for i in idx:
        assert torch.floor(torch.log2(x[i:i + 1])).item() == fl[i], (hex(u[i]), fl[i])
    exact = (np.frexp(x.numpy().astype(np.float64))[1] - 1).astype(np.float32)
    print("values", len(u), "bumped above the exact exponent:", int((fl != exact).sum()))
    assert set(np.unique(fl - exact)) <= {0.0, 1.0}
    np.savez_compressed(os.path.join(HERE, "log2_f32.npz"), bits=u, floor_log2=fl.astype(np.int16))


if __name__ == "__main__":
    main()
