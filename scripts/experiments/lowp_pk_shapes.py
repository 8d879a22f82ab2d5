#!/usr/bin/env python3
"""Random shapes / axes / blocks through the packed fake-quant kernels (in-dtype and float32 semantics) against the kernels they replace
(MSQ_OUTLIER_LOWP_PK = 0): 1 ... 4 dimensions, extents that are and are not multiples of the block, odd trailing extents (no column pairs),
single-block tensors, partial last waves.  python scripts/experiments/lowp_pk_shapes.py [cases]"""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import lowp_pk_fuzz as F

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    rng = random.Random(11)
    g = torch.Generator(device=F.dev).manual_seed(11)
    bad = 0
    for it in range(n):
        nd = rng.randint(1, 4)
        shape = [rng.choice([1, 2, 3, 5, 8, 16, 24, 32, 40, 64, 96, 128, 130]) for _ in range(nd)]
        axis = rng.randrange(nd)
        bs = rng.choice([8, 16, 32, 64])
        if rng.random() < 0.7:
            shape[axis] = bs * rng.randint(1, 6)
        if torch.tensor(shape).prod().item() > 2_000_000:
            continue
        dt = rng.choice([torch.float16, torch.bfloat16])
        W = (torch.randn(*shape, generator=g, device=F.dev) * rng.choice([0.02, 1.0, 30.0])).to(dt)
        fi, fo = rng.choice(F.COMBOS)
        native = rng.random() < 0.5
        sd = rng.choice([1.0, 2.0, 3.0])
        a = F.run(W, fi, fo, sd, axis, bs, 8, 1, ("s", "s"), native)
        b = F.run(W, fi, fo, sd, axis, bs, 8, 0, None, native)
        ok = F.same(a["out"], b["out"]) and torch.equal(a["mask"], b["mask"]) and F.same(a["e_in"], b["e_in"]) and F.same(a["e_out"], b["e_out"]) and a["status"] == b["status"]
        if not ok:
            bad += 1
            print("MISMATCH", shape, axis, bs, str(dt)[6:], fi, fo, "native" if native else "float32 semantics", sd)
    h, w = F.HANDED.get(("s", "s"), (0, 1))
    print("cases %d, mismatching %d; waves handed back %d of %d" % (n, bad, h, w))
    return 1 if bad else 0

if __name__ == "__main__":
    sys.exit(main())
