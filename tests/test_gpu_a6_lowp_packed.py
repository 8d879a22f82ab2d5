"""SURVEY 8 row a6, fp16 / bf16 tensors computed in their dtype (llm/llama.py:238 -> utils/quant.py:147-266 on a Half tensor): the PACKED kernels
k_outlier_lowp_pk / _pk2 (csrc/msq_quant_lowp.hip, round 6).  What pins them:
  * the reference-made goldens (tests/golden/outlier_lowp.npz) -- the cases the packed kernels take are counted here, the comparison itself is
    test_lowp_outlier_fakequant_golden_gpu (default path = packed kernels since round 6);
  * the oracle on 16.8 M weights (test_lowp_llama_sized_weight_vs_oracle, same remark);
  * the op-by-op kernel k_outlier_lowp (tuning key MSQ_OUTLIER_LOWP_PK = 0), itself pinned by the two above, on adversarial inputs: values one and
    two ulps either side of every tie / grid point, the excepted magnitude of DESIGN.md 5.002, scales through the whole exponent range, fp16
    subnormals, blocks without outliers / without inliers, constant blocks, zero bounds, NaN / Inf members -- values, masks, both exponents and
    the status word bit for bit (scripts/experiments/lowp_pk_fuzz.py is the long form of the same generator)."""
import json
import os
import sys

import numpy as np
import pytest
import torch

from gpu_common import G, ROOT, dev

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(ROOT, "scripts", "experiments"))


def _fuzz():
    import lowp_pk_fuzz as F
    return F


@pytest.mark.parametrize("dn", ["f16", "bf16"])
def test_packed_kernels_take_a_weight_and_equal_the_op_by_op_kernel(msq, dn):
    """A Llama-sized projection in the checkpoint dtype, the harness call (int2 / fp4, blocks of 16 along out_features) and the BASELINE pair:
    fewer than 1 % of the waves are handed back, and the tensor, the mask and the exponents are the op-by-op kernel's."""
    F = _fuzz()
    dt = torch.float16 if dn == "f16" else torch.bfloat16
    g = torch.Generator(device=dev()).manual_seed(5)
    W = torch.randn(4096, 4096, generator=g, device=dev()) * 0.02
    W[torch.rand(4096, 4096, generator=g, device=dev()) < 0.005] *= 16
    W = W.to(dt)
    for fi, fo, ax, bs in (("int2", "fp4", 0, 16), ("fp4_e2m1", "fp8_e4m3", -1, 32), ("fp4_e2m1", "fp8_e4m3", 0, 32), ("int2", "fp4", -1, 16)):
        F.HANDED.clear()
        a = F.run(W, fi, fo, 2.0, ax, bs, 8, 1, tag=("w", "w"))
        b = F.run(W, fi, fo, 2.0, ax, bs, 8, 0)
        h, w = F.HANDED[("w", "w")]
        assert h < 0.01 * w, (fi, fo, ax, bs, h, w)
        assert F.same(a["out"], b["out"]) and torch.equal(a["mask"], b["mask"]) and F.same(a["e_in"], b["e_in"]) and F.same(a["e_out"], b["e_out"])
        assert a["status"] == b["status"]


def test_packed_kernels_without_workspace_fall_back(msq):
    """No workspace (a C caller of the round-5 ABI): the op-by-op kernel alone, same result; the tuning key switches likewise."""
    from msq._lib import ptr, check, current_stream, lib
    from msq.formats import format_id
    L = lib()
    W = (torch.randn(256, 512, device=dev()) * 0.02).half()
    ref = msq.quant.outlier_fakequant(W, 8, 8, "int2", "fp4", 2, 0, 16)["out"]
    out = torch.empty_like(W)
    check(L.msq_outlier_fakequant(ptr(W), ptr(out), None, None, None, None, None, None, 0, 0x11, 1, 256, 512, 16, format_id("int2"), format_id("fp4"),
                                  8, 8, 2.0, 0, 0, 0, current_stream(dev())), "msq_outlier_fakequant")
    assert torch.equal(out.view(torch.int16), ref.view(torch.int16))
    assert L.msq_set_tuning(b"MSQ_OUTLIER_LOWP_PK", 0) == 0
    try:
        assert torch.equal(msq.quant.outlier_fakequant(W, 8, 8, "int2", "fp4", 2, 0, 16)["out"].view(torch.int16), ref.view(torch.int16))
    finally:
        L.msq_set_tuning(b"MSQ_OUTLIER_LOWP_PK", 1)
    assert L.msq_outlier_workspace_bytes(1, 256, 512, 16, 0) >= 64 + 8 * (16 * 512 // 64)


def test_packed_vs_op_by_op_adversarial(msq):
    """The generator of scripts/experiments/lowp_pk_fuzz.py, one round on the two small shapes: every kind of input, five format pairs, both axes,
    blocks of 8 / 16 / 32 / 64, three (std_dev, scale bits) settings, both dtypes -- and every kind is partly ON the packed path (a comparison of the
    op-by-op kernel with itself proves nothing)."""
    F = _fuzz()
    F.HANDED.clear()
    g = torch.Generator(device=dev()).manual_seed(99)
    kinds = ["weights", "scales", "ties", "tiesc", "negative", "positive", "sparse", "constant", "subnormal", "special"]
    n = 0
    for dt in (torch.float16, torch.bfloat16):
        for kind in kinds:
            for shape in ((256, 512), (96, 160)):
                W = F.make(kind, shape, dt, g)
                for fi, fo in F.COMBOS:
                    for axis, bs in ((0, 16), (-1, 32), (0, 32), (-1, 16), (0, 8), (-1, 8), (0, 64), (-1, 64)):
                        if W.shape[axis] % bs:
                            continue
                        for sd, sb in ((2.0, 8), (3.0, 8), (1.0, 4)):
                            a = F.run(W, fi, fo, sd, axis, bs, sb, 1, (str(dt)[6:], kind))
                            b = F.run(W, fi, fo, sd, axis, bs, sb, 0)
                            assert F.same(a["out"], b["out"]), (dt, kind, shape, fi, fo, axis, bs, sd, sb)
                            assert torch.equal(a["mask"], b["mask"]) and F.same(a["e_in"], b["e_in"]) and F.same(a["e_out"], b["e_out"]), (dt, kind, shape, fi, fo, axis, bs)
                            assert a["status"] == b["status"], (dt, kind, shape, fi, fo, axis, bs, sd, sb)
                            n += 1
    assert n == 2 * len(kinds) * (8 + 6) * 5 * 3          # [256, 512]: all eight (axis, block) settings; [96, 160]: 64 does not divide either axis
    for (dn, kind), (h, w) in F.HANDED.items():
        assert h < w, (dn, kind, h, w)                     # some waves of every kind stayed on the packed path
    for dn in ("float16", "bfloat16"):
        h, w = F.HANDED[(dn, "ties")]
        assert h < 0.5 * w, (dn, h, w)                     # the tie / excepted-magnitude inputs mostly did


def test_goldens_that_run_on_the_packed_kernels(msq):
    """How many of the reference-made half-tensor cases (outlier_lowp.npz) the packed kernels take whole (no wave handed back): those goldens pin the
    packed arithmetic directly.  Compared here again, through the C ABI with a workspace of our own."""
    F = _fuzz()
    z = np.load(os.path.join(G, "outlier_lowp.npz"))
    meta = json.load(open(os.path.join(G, "outlier_lowp_meta.json")))
    pairs = {("int2", "fp4"), ("int2", "fp4_e2m1"), ("fp4_e2m1", "fp8_e4m3"), ("fp4_e2m1", "fp4_e2m1"), ("fp4_e2m1", "fp8_e5m2"), ("fp8_e4m3", "fp8_e4m3"),
             ("fp4", "fp8_e4m3")}
    whole = taken = 0
    for key, m in sorted(meta.items()):
        dn, tname, cname = key.split("|")
        isb, osb, fi, fo, sd, axes, bs, rnd = m["cfg"]
        if "assert" in m or rnd != "nearest" or (fi, fo) not in pairs or bs not in (8, 16, 32, 64) or isb != osb:
            continue
        bits = torch.from_numpy(z[f"in|{dn}|{tname}"].astype(np.int16))
        A = bits.view(torch.float16 if dn == "f16" else torch.bfloat16).to(dev())
        ax = axes[0] % A.ndim
        if A.shape[ax] % bs:
            continue
        F.HANDED.clear()
        r = F.run(A, fi, fo, sd, ax, bs, isb, 1, tag=("g", "g"))
        h, w = F.HANDED[("g", "g")]
        got = r["out"].view(torch.int16).cpu().numpy().view(np.uint16)
        ref = z[f"out|{key}"]
        f = lambda u: torch.from_numpy(u.astype(np.int16)).view(A.dtype).float().numpy()
        assert ((got == ref) | (np.isnan(f(got)) & np.isnan(f(ref)))).all(), key
        taken += 1
        whole += int(h == 0)
    assert taken >= 40 and whole >= 20, (taken, whole)


def test_packed_kernels_on_tensors_with_leading_and_trailing_dims(msq):
    """[pre, axis, post] with pre > 1 and post > 1 (the KV-cache shape [B, H, T, D] quantised along T or D; a stacked-expert weight): the strided
    kernel walks (p, block, column pair), the contiguous one p * nblk blocks; partial last waves on both; odd post (no column pairs) and a ragged
    axis take the op-by-op kernel."""
    F = _fuzz()
    g = torch.Generator(device=dev()).manual_seed(3)
    keep, msq.quant.CHECK_NAN = msq.quant.CHECK_NAN, False        # (compare the tensors themselves: the reference's NaN blocks included)
    try:
        _nd_cases(msq, F, g)
    finally:
        msq.quant.CHECK_NAN = keep


def _nd_cases(msq, F, g):
    for dt in (torch.float16, torch.bfloat16):
        for shape in ((6, 64, 40), (3, 5, 96, 32), (2, 32, 7), (4, 40, 64)):
            W = (torch.randn(*shape, generator=g, device=dev()) * 0.05).to(dt)
            for axis in range(1, len(shape)):
                for bs in (16, 32):
                    for fi, fo in (("int2", "fp4"), ("fp4_e2m1", "fp8_e4m3")):
                        a = msq.quant.outlier_fakequant(W, 8, 8, fi, fo, 2, axis, bs, want_mask=True, want_exps=True)
                        assert msq._lib.lib().msq_set_tuning(b"MSQ_OUTLIER_LOWP_PK", 0) == 0
                        try:
                            b = msq.quant.outlier_fakequant(W, 8, 8, fi, fo, 2, axis, bs, want_mask=True, want_exps=True)
                        finally:
                            msq._lib.lib().msq_set_tuning(b"MSQ_OUTLIER_LOWP_PK", 1)
                        assert F.same(a["out"], b["out"]) and torch.equal(a["mask"], b["mask"]), (dt, shape, axis, bs, fi)
                        assert F.same(a["e_in"], b["e_in"]) and F.same(a["e_out"], b["e_out"]), (dt, shape, axis, bs, fi)


def test_float32_semantics_on_the_packed_kernels(msq):
    """fp16 / bf16 tensors computed in FLOAT32 (dtype 1 / 2 of msq_outlier_fakequant: what the MicroScopiQ KV cache runs) on the packed kernels
    (outlier_block_pk32) against the float32 kernels (MSQ_OUTLIER_LOWP_PK = 0 -> k_outlier_contig / _strided with 16-bit loads and stores), bit
    for bit: values, masks, exponents, status -- on the adversarial generator, and on a KV-cache-like tensor where (almost) every wave must stay
    on the packed path."""
    F = _fuzz()
    F.HANDED.clear()
    g = torch.Generator(device=dev()).manual_seed(123)
    kinds = ["weights", "scales", "ties", "tiesc", "negative", "positive", "sparse", "constant", "subnormal", "special"]
    for dt in (torch.float16, torch.bfloat16):
        for kind in kinds:
            W = F.make(kind, (256, 512), dt, g)
            for fi, fo in F.COMBOS:
                for axis, bs in ((0, 16), (-1, 32), (0, 32), (-1, 16), (0, 8), (-1, 8)):
                    for sd, sb in ((2.0, 8), (1.0, 4)):
                        a = F.run(W, fi, fo, sd, axis, bs, sb, 1, (str(dt)[6:], kind), native=False)
                        b = F.run(W, fi, fo, sd, axis, bs, sb, 0, None, native=False)
                        assert F.same(a["out"], b["out"]), (dt, kind, fi, fo, axis, bs, sd, sb)
                        assert torch.equal(a["mask"], b["mask"]) and F.same(a["e_in"], b["e_in"]) and F.same(a["e_out"], b["e_out"]), (dt, kind, fi, fo, axis, bs)
                        assert a["status"] == b["status"], (dt, kind, fi, fo, axis, bs, sd, sb)
    for (dn, kind), (h, w) in F.HANDED.items():
        if kind not in ("scales", "special"):
            assert h < w, (dn, kind, h, w)
    # a layer cache [1, 8, 1024, 128], keys (blocks along tokens) and values (blocks along head_dim): the method of kvcache.mx_quantize_*
    for dt in (torch.float16, torch.bfloat16):
        kv = torch.randn(1, 8, 1024, 128, generator=g, device=dev()).to(dt)
        for axis in (2, 3):
            F.HANDED.clear()
            a = F.run(kv, "fp4_e2m1", "fp8_e4m3", 2.0, axis, 32, 8, 1, ("kv", "kv"), native=False)
            b = F.run(kv, "fp4_e2m1", "fp8_e4m3", 2.0, axis, 32, 8, 0, None, native=False)
            h, w = F.HANDED[("kv", "kv")]
            assert h <= 0.01 * w, (dt, axis, h, w)
            assert F.same(a["out"], b["out"]) and torch.equal(a["mask"], b["mask"]) and F.same(a["e_in"], b["e_in"]) and F.same(a["e_out"], b["e_out"])
            ref = msq.quant.outlier_fakequant(kv.float(), 8, 8, "fp4_e2m1", "fp8_e4m3", 2, axis, 32)["out"].to(dt)      # upcast, float32 kernel, downcast
            assert torch.equal(a["out"].view(torch.int16), ref.view(torch.int16))
