#!/usr/bin/env python3
"""Randomised differential run (round 5) of the kernels this round added or rewrote, against the CPU oracle:
  (a) the KV-cache MX quantiser in the cache dtype -- keys (k_mx_lowp_pair4 / k_mx_lowp_pair) and values (k_mx_lowp_vec): random [B, H, S, D],
      fp16 / bf16, element formats, magnitudes from subnormal to near-overflow, planted zeros / Inf / NaN;  bit for bit.
  (b) the activation producers -- RMSNorm, simd_mul bit for bit; silu x up up to the device expf (counted, <= 3 per case tolerated);
      fused packs byte-equal to producer + packer.
Seeded; prints failures only + a summary.  Usage: python scripts/experiments/fuzz_kv_producers.py [cases] [seed]"""
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

import msq
from msq import kvcache, qlinear, vector_ops as V
from msq._lib import lib
from oracle import oracle as O

dev = torch.device("cuda:0")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 11)
g = torch.Generator().manual_seed(rnd.randrange(1 << 30))


def eq_bits(a, b):
    a = np.ascontiguousarray(a, np.float32); b = np.ascontiguousarray(b, np.float32)
    return bool(((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))).all())


bad = 0
# ---- (a) KV
for case in range(cases):
    dtype = rnd.choice([torch.float16, torch.bfloat16]); dn = "f16" if dtype == torch.float16 else "bf16"
    B, H = rnd.choice([1, 2]), rnd.choice([1, 3, 8])
    S = rnd.choice([32, 64, 70, 96, 128, 33, 256])
    D = rnd.choice([64, 80, 96, 128, 160, 256])
    fmt = rnd.choice(["fp8_e4m3", "fp8_e4m3", "fp8_e4m3", "fp8_e5m2", "fp6_e3m2", "fp6_e2m3", "fp4_e2m1", "int8", "int4"])
    scale = 2.0 ** rnd.choice([-20, -14, -9, -6, -3, 0, 0, 0, 3, 8, 12] if dtype == torch.float16 else [-120, -60, -20, -9, -3, 0, 0, 0, 3, 20, 60, 110])
    x = (torch.randn(B, H, S, D, generator=g) * scale)
    x[torch.rand(B, H, S, D, generator=g) < 0.1] *= 2.0 ** -7
    x = x.to(dtype)
    if rnd.random() < 0.3:
        x[0, 0, :min(S, 32), :min(D, 32)] = 0
    if rnd.random() < 0.2:
        x[0, 0, rnd.randrange(S), rnd.randrange(D)] = float("inf")
    if rnd.random() < 0.2:
        x[-1, -1, rnd.randrange(S), rnd.randrange(D)] = float("nan")
    xd = x.to(dev)
    pair4 = rnd.choice([0, 1])
    lib().msq_set_tuning(b"MSQ_MX_LOWP_PAIR4", pair4)
    try:
        kq = kvcache.mx_quantize_keys(xd, fmt, 32).float().cpu().numpy()
        vq = kvcache.mx_quantize_values(xd, fmt, 32).float().cpu().numpy() if D % 32 == 0 else None
    finally:
        lib().msq_set_tuning(b"MSQ_MX_LOWP_PAIR4", 1)
    ko = O.quantize_mx_lowp(x.float().numpy(), dn, 8, fmt, 2, 32)
    ok = eq_bits(kq, ko)
    if vq is not None:
        ok = ok and eq_bits(vq, O.quantize_mx_lowp(x.float().numpy(), dn, 8, fmt, 3, 32))
    if not ok:
        bad += 1
        print("KV FAIL", dn, (B, H, S, D), fmt, "scale 2^%d" % int(np.log2(scale)), "pair4", pair4, flush=True)
print("KV cases", cases, "failures", bad, flush=True)

# ---- (b) producers
bad_p = 0; silu_off = 0
mn16 = 3.3895313892515355e38
for case in range(cases):
    bf = rnd.choice([16, 16, 12])
    rd = "nearest" if bf == 16 else rnd.choice(["nearest", "even", "floor"])
    specs = msq.specs.finalize_mx_specs({"w_elem_format": "fp4_e2m1", "a_elem_format": "fp8_e4m3", "scale_bits": 8, "block_size": 32, "custom_cuda": True,
                                         "bfloat": bf, "round": rd})
    bits = bf - 7
    mn = 2.0 ** 127 * (2 ** (bits - 1) - 1) / 2 ** (bits - 2)
    H = rnd.choice([128, 200, 384, 512, 1024, 1536, 4096, 8192, 520])
    rows = rnd.choice([1, 3, 17, 64])
    sc = 2.0 ** rnd.choice([-30, -8, 0, 0, 4, 30])
    x = torch.randn(rows, H, generator=g) * sc
    w = torch.randn(H, generator=g) * 0.5 + 1
    b = torch.randn(H, generator=g) * 0.1 if rnd.random() < 0.5 else None
    eps = rnd.choice([1e-6, 1e-5, 1e-12])
    y = V.rms_norm(x.to(dev), w.to(dev), None if b is None else b.to(dev), eps, specs).cpu().numpy()
    yo = O.vec_rmsnorm(x.numpy(), w.numpy(), np.zeros(H, np.float32) if b is None else b.numpy(), eps, bits, 8, mn, rd)
    ok = eq_bits(y, yo)
    why = []
    if not ok:
        dm = np.ascontiguousarray(y).view(np.uint32) != np.ascontiguousarray(yo).view(np.uint32)
        i = tuple(np.argwhere(dm)[0])
        why.append("rmsnorm: %d values differ, first %r vs oracle %r (x %r)" % (int(dm.sum()), float(y[i]), float(yo[i]), float(x[i])))
    if H % 128 == 0 and rd == "nearest":
        c0, s0 = qlinear.mx_pack_act(torch.from_numpy(y).to(dev))
        c1, s1 = V.rms_norm_mx_pack(x.to(dev), w.to(dev), None if b is None else b.to(dev), eps, specs)
        ok = ok and torch.equal(c0, c1) and torch.equal(s0, s1)
        xb = x.to(torch.bfloat16).to(dev)
        c2, s2 = V.rms_norm_mx_pack(xb, w.to(dev), None if b is None else b.to(dev), eps, specs)
        c3, s3 = V.rms_norm_mx_pack(xb.float(), w.to(dev), None if b is None else b.to(dev), eps, specs)
        ok = ok and torch.equal(c2, c3) and torch.equal(s2, s3)
    I = rnd.choice([128, 256, 1408, 2048, 11008, 200])
    M = rnd.choice([1, 5, 40])
    gu = torch.randn(M, 2 * I, generator=g) * rnd.choice([0.1, 1.0, 3.0, 20.0, 60.0])
    gate, up = gu[:, :I], gu[:, I:]
    m = V.silu_mul(gate.to(dev), up.to(dev), specs).cpu().numpy()
    mo = O.vec_mul(O.vec_silu(gate.contiguous().numpy(), bits, 8, mn, rd), up.contiguous().numpy(), bits, 8, mn, rd)
    nd = int(((m != mo) & ~(np.isnan(m) & np.isnan(mo))).sum())
    silu_off += nd
    ok = ok and nd <= 3
    mm = V.simd_mul(gate.to(dev), up.to(dev), mx_specs=specs).cpu().numpy()
    mmo = O.vec_mul(gate.contiguous().numpy(), up.contiguous().numpy(), bits, 8, mn, rd)
    if not eq_bits(mm, mmo):
        i = tuple(np.argwhere(mm.view(np.uint32) != mmo.view(np.uint32))[0])
        why.append("simd_mul: %r x %r -> %r vs oracle %r" % (float(gate[i]), float(up[i]), float(mm[i]), float(mmo[i])))
        ok = False
    if I % 128 == 0 and rd == "nearest":
        gud = gu.to(dev)
        c0, s0 = qlinear.mx_pack_act(torch.from_numpy(m).to(dev))
        c1, s1 = V.silu_mul(gud[:, :I], gud[:, I:], specs, pack=True)
        ok = ok and torch.equal(c0, c1) and torch.equal(s0, s1)
    if not ok:
        bad_p += 1
        print("PRODUCER FAIL bfloat", bf, rd, "H", H, "rows", rows, "I", I, "M", M, "silu diffs", nd, why, flush=True)
print("producer cases", cases, "failures", bad_p, "silu x up elements off by the device expf:", silu_off, flush=True)
print("TOTAL failures", bad + bad_p)
sys.exit(1 if bad + bad_p else 0)
