#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ by IMPORTING the reference.

Run in the build container only (needs /root/reference; the GPU box has none):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The reference is copied to a scratch directory first so that importing it can
never write into /root/reference.  Only DATA (inputs + the reference's outputs)
is stored; no reference source is copied into the repository.

The KAT vectors in kat_vectors.json are the input/expected arrays of the
reference's own known-answer tests (number_system/mx/tests/*), transcribed as
data, each tagged with the test it comes from.
"""
import json
import os
import shutil
import sys
import tempfile
import warnings

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True
warnings.filterwarnings("ignore")

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def _import_reference():
    scratch = tempfile.mkdtemp(prefix="msq_refcopy_")
    for d in ("utils", "number_system"):
        shutil.copytree(os.path.join(REF, d), os.path.join(scratch, d))
    sys.path[:0] = [os.path.join(scratch, "number_system"), os.path.join(scratch, "number_system", "posit"), scratch]
    import mx  # noqa
    from utils import quant  # noqa
    import importlib
    from mx import mx_ops, elemwise_ops, formats, specs  # noqa
    linear = importlib.import_module("mx.linear")  # mx.linear attr is shadowed by the function
    import Posit as posit_mod  # noqa
    return scratch, quant, mx_ops, elemwise_ops, formats, linear, specs, posit_mod


def sweep_inputs():
    """fp32 probe values for the scalar codec: +-(1+j/64)*2^e, specials, subnormals."""
    v = []
    for e in range(-13, 11):
        for j in range(64):
            v.append((1.0 + j / 64.0) * 2.0 ** e)
    v = np.array(v, dtype=np.float32)
    rng = np.random.RandomState(0)
    r = (rng.randn(1024) * np.exp2(rng.randint(-10, 9, 1024))).astype(np.float32)
    sp = np.array([0.0, 448.0, 449.0, 464.0, 480.0, 57344.0, 61440.0, 65504.0, 65520.0, 1e30, 2.0 ** -126,
                   2.0 ** -127, 2.0 ** -140, 2.0 ** -149, 3.0 * 2.0 ** -130, 6.0, 6.5, 7.0, 7.5, 28.0, 30.0,
                   1.984375, 1.9921875, 1.75, 1.875, 0.9999999, 1.0000001], dtype=np.float32)
    x = np.concatenate([v, -v, r, sp, -sp]).astype(np.float32)
    return x


def tensors():
    """Seeded inputs for the block quantizers (small)."""
    g = torch.Generator().manual_seed(1234)
    t = {}
    w = torch.randn(64, 96, generator=g) * 0.02
    t["gauss"] = w
    h = torch.randn(64, 96, generator=g) * 0.02
    idx = torch.rand(64, 96, generator=g) < 0.01
    h[idx] *= 20.0
    t["heavy"] = h
    z = torch.randn(48, 64, generator=g) * 0.05
    z[0:16, 3] = 0.0          # an all-zero block along axis 0 (bs 16)
    z[5, 0:32] = 0.0          # an all-zero block along axis -1 (bs 32)
    z[16:32, 7] = -z[16:32, 7].abs()  # an all-negative block
    z[20, 32:64] = z[20, 32:64].abs()  # an all-positive block
    t["zeros"] = z
    t["ragged"] = torch.randn(20, 40, generator=g) * 0.1          # 20 % 16 != 0, 40 % 32 != 0
    t["fp16rep"] = (torch.randn(32, 64, generator=g) * 0.03).half().float()
    t["big"] = torch.randn(32, 32, generator=g) * 3000.0
    t["tiny"] = torch.randn(32, 32, generator=g) * 1e-30
    t["act3d"] = torch.randn(2, 40, 64, generator=g)
    return t


def kat_vectors():
    """Input / expected arrays of the reference's own known-answer tests, as data.
    Each entry: source test, the call it makes, x, expected t.  inf in `t` means
    'Inf or NaN' exactly as the reference's check_diff_quantize(handle_infs=True)
    treats them (number_system/mx/tests/common_lib.py:91-134)."""
    inf = float("inf")
    K = []
    # number_system/mx/tests/test_corners_mx.py:64-79  test_mx_rounding
    K.append(dict(src="test_corners_mx.py:64-79 test_mx_rounding", fn="quantize_mx", scale_bits=8, fmt="int4",
                  block_size=0, axis=1, round="nearest",
                  x=[[-1.0, -1.75, -1.99, 1.99], [1.0, -1.75, -1.99, 1.99]],
                  t=[[-1.0, -1.75, -1.75, 1.75], [1.0, -1.75, -1.75, 1.75]]))
    # test_corners_mx.py:82-124  test_mx_hw_test
    K.append(dict(src="test_corners_mx.py:82-124 test_mx_hw_test", fn="quantize_mx", scale_bits=8, fmt="int8",
                  block_size=10, axis=1, round="nearest",
                  x=[[1.0] * 10,
                     [1.0] * 5 + [2.0] * 5,
                     [-1.0] * 5 + [-2.0] * 5,
                     [1.0] * 5 + [-2.0] * 5,
                     [1.015625, 1.0234375, 1.03125, 1.0390625, 1.25, 1.2578125, 1.9375, 1.9453125, 1.984375, 1.9921875],
                     [-1.984375, -1.9765625, -1.96875, -1.9609375, -1.9375, -1.9296875, -1.75, -1.7421875, -1.0, -1.9921875],
                     [1.99609375, 1.98828125, 0.0, 0.00390625, 0.0078125, 0.01171875, -0.015625, -0.01171875, -0.0078125, -0.00390625]],
                  t=[[1.0] * 10,
                     [1.0] * 5 + [2.0] * 5,
                     [-1.0] * 5 + [-2.0] * 5,
                     [1.0] * 5 + [-2.0] * 5,
                     [1.015625, 1.03125, 1.03125, 1.046875, 1.25, 1.265625, 1.9375, 1.953125, 1.984375, 1.984375],
                     [-1.984375, -1.984375, -1.96875, -1.96875, -1.9375, -1.9375, -1.75, -1.75, -1.0, -1.984375],
                     [1.984375, 1.984375, 0.0, 0.0, 0.015625, 0.015625, -0.015625, -0.015625, -0.015625, 0.0]]))
    # test_corners_mx.py:27-60 test_mx_nans (val in NaN, Inf, -Inf; fmt in fp8_e4m3, fp4_e2m1, int4; round floor)
    for val in ("nan", "inf", "-inf"):
        for fmt in ("fp8_e4m3", "fp4_e2m1", "int4"):
            K.append(dict(src="test_corners_mx.py:27-60 test_mx_nans", fn="quantize_mx", scale_bits=8, fmt=fmt,
                          block_size=4, axis=-1, round="floor", x=[val, 0, 2.0 ** 127, 0], t=["nan"] * 4))
    # test_fp8_e4m3_fix.py:17-31
    K.append(dict(src="test_fp8_e4m3_fix.py:17-31 test_fp8_e4m3_fix_pytorch", fn="quantize_mx", scale_bits=8,
                  fmt="fp8_e4m3", block_size=8, axis=-1, round="nearest",
                  x=[485, 475, 400, -485, -475, -450, 2.0 ** -6, -(2.0 ** -6)],
                  t=[448, 448, 416, -448, -448, -448, 2.0 ** -6, -(2.0 ** -6)]))
    # test_fp8_e4m3_fix.py:87-119 test_mxfp8_e4m3_round (round in nearest, even)
    x_ = [-0.582733273506, -0.256973713636, 0.506033003330, 0.400039970875, -0.437093466520, 1.275019764900,
          -2.123294353485, 1.514909625053, -2.660086154938, -0.200791791081, -0.060985822231, -0.209203109145,
          2.385987281799, 0.062245476991, 0.217003762722, -0.857734560966, 0.507945835590, 0.896152675152,
          -0.751049160957, -0.488164335489, 0.805381953716, -0.172028362751, 0.271137863398, -0.503807783127,
          1.879478693008, -0.294227510691, 0.968807995319, 0.670037031174, -0.871595799923, 0.304561793804,
          -0.567594051361, -1.265962004662]
    t_ = [-0.5625, -0.25, 0.5, 0.40625, -0.4375, 1.25, -2.0, 1.5, -2.75, -0.203125, -0.0625, -0.203125, 2.5, 0.0625,
          0.21875, -0.875, 0.5, 0.875, -0.75, -0.5, 0.8125, -0.171875, 0.28125, -0.5, 1.875, -0.28125, 1.0, 0.6875,
          -0.875, 0.3125, -0.5625, -1.25]
    for rnd in ("nearest", "even"):
        K.append(dict(src="test_fp8_e4m3_fix.py:87-119 test_mxfp8_e4m3_round", fn="quantize_mx", scale_bits=8,
                      fmt="fp8_e4m3", block_size=32, axis=0, round=rnd, x=x_, t=t_))
    # test_e5m0_scale.py:23-45 (scale_bits=5, int2, also on -x / -t)
    xs = [2.0 ** 16, 2.0 ** 15, 2.0 ** -16, 2.0 ** -17, 2.0 ** -15, 2.0 ** -14, 1, 0]
    tsv = ["inf", 2.0 ** 15, 2.0 ** -15, 0, 2.0 ** -15, 2.0 ** -14, 1, 0]
    K.append(dict(src="test_e5m0_scale.py:23-45 test_e5m0_scale_pytorch", fn="quantize_mx", scale_bits=5, fmt="int2",
                  block_size=8, axis=-1, round="nearest", negate_too=True,
                  x=[[e] + 7 * [0] for e in xs], t=[[e] + 7 * [("inf" if e == "inf" else 0)] for e in tsv]))
    # test_corners_elemwise.py (bfloat / fp scalar codec corners)
    E = []
    E.append(dict(src="test_corners_elemwise.py:155-163 test_bfloat16_round", bits=9, exp_bits=8, round="nearest",
                  saturate=False, allow_denorm=True, max_norm="bf16",
                  x=[65535., -65535., -1.9985847, 0.4999000132083893, -0.49968934059143066],
                  t=[65536., -65536., -2.0, 0.5, -0.5]))
    for dn in (False, True):
        E.append(dict(src="test_corners_elemwise.py:166-189 test_float16_subnorms", bits=12, exp_bits=5, round="nearest",
                      saturate=False, allow_denorm=dn, max_norm="fp16",
                      x=[2.0 ** -14, 2.0 ** -14 * 0.5, 2.0 ** -14 * 2.0 ** -10, 2.0 ** -14 * 2.0 ** -11, 2.0 ** -14 * 2.0 ** -12,
                         2.0 ** -14 * (1023 / 1024), 2.0 ** -14 * (2047 / 2048)],
                      t=([2.0 ** -14, 2.0 ** -14 * 0.5, 2.0 ** -14 * 2.0 ** -10, 2.0 ** -14 * 2.0 ** -10, 0,
                          2.0 ** -14 * (1023 / 1024), 2.0 ** -14] if dn else [2.0 ** -14, 0, 0, 0, 0, 0, 0])))
        E.append(dict(src="test_corners_elemwise.py:207-229 test_bfloat16_subnorms", bits=9, exp_bits=8, round="nearest",
                      saturate=False, allow_denorm=dn, max_norm="bf16",
                      x=[2.0 ** -126, 2.0 ** -126 * 0.5, 2.0 ** -126 * 2.0 ** -7, 2.0 ** -126 * 2.0 ** -8, 2.0 ** -126 * 2.0 ** -9,
                         2.0 ** -126 * (127 / 128), 2.0 ** -126 * (255 / 256)],
                      t=([2.0 ** -126, 2.0 ** -126 * 0.5, 2.0 ** -126 * 2.0 ** -7, 2.0 ** -126 * 2.0 ** -7, 0,
                          2.0 ** -126 * (127 / 128), 2.0 ** -126] if dn else [2.0 ** -126, 0, 0, 0, 0, 0, 0])))
    E.append(dict(src="test_corners_elemwise.py:192-204 test_bfloat16_limits", bits=9, exp_bits=8, round="nearest",
                  saturate=False, allow_denorm=True, max_norm="bf16",
                  x=[2.0 ** 127 * 1.9921875, 2.0 ** 127 * 1.9921874, -2.0 ** 127 * 1.9921875, 2.0 ** 127 * 1.99609375,
                     -2.0 ** 127 * 1.99609375],
                  t=[2.0 ** 127 * 1.9921875, 2.0 ** 127 * 1.9921875, -2.0 ** 127 * 1.9921875, "inf", "-inf"]))
    E.append(dict(src="test_corners_elemwise.py:232-243 test_subnorm_rne", bits=9, exp_bits=8, round="even",
                  saturate=False, allow_denorm=True, max_norm="bf16",
                  x=[2.0 ** -126 * (253 / 256), 2.0 ** -126 * (251 / 256)],
                  t=[2.0 ** -126 * (126 / 128), 2.0 ** -126 * (126 / 128)]))
    E.append(dict(src="test_corners_elemwise.py:58-84 test_fp16_max", bits=12, exp_bits=5, round="nearest",
                  saturate=False, allow_denorm=True, max_norm="fp16",
                  x=[65504., -65504., 65519., -65519., 65520., -65520.],
                  t=[65504., -65504., 65504., -65504., "inf", "-inf"]))
    return {"mx": K, "elemwise": E}


CFGS = [
    # name, in_sb, out_sb, in_fmt, out_fmt, std_dev, axes, bs, round
    ("int2_fp4_bs16_ax0", 8, 8, "int2", "fp4", 2, [0], 16, "nearest"),          # llm/llama.py:230-237
    ("fp4_fp8e4m3_bs32_ax0", 8, 8, "fp4_e2m1", "fp8_e4m3", 2, [0], 32, "nearest"),
    ("fp4_fp8e4m3_bs32_axm1", 8, 8, "fp4_e2m1", "fp8_e4m3", 2, [-1], 32, "nearest"),
    ("fp6e3m2_fp8e4m3_bs32_axm1", 8, 8, "fp6_e3m2", "fp8_e4m3", 2, [-1], 32, "nearest"),
    ("int4_int8_bs16_ax0", 8, 8, "int4", "int8", 2, [0], 16, "nearest"),
    ("fp4_fp8e5m2_bs16_axm1", 8, 8, "fp4", "fp8_e5m2", 3, [-1], 16, "nearest"),
    ("fp4_fp8e4m3_bs32_axm1_even", 8, 8, "fp4_e2m1", "fp8_e4m3", 2, [-1], 32, "even"),
    ("fp4_fp8e4m3_bs32_ax0_floor", 8, 8, "fp4_e2m1", "fp8_e4m3", 2, [0], 32, "floor"),
    ("fp4_fp8e4m3_bs32_axm1_sd1p5", 8, 8, "fp4_e2m1", "fp8_e4m3", 1.5, [-1], 32, "nearest"),
    ("fp6e2m3_fp8e4m3_bs8_ax0_sb45", 4, 5, "fp6_e2m3", "fp8_e4m3", 2, [0], 8, "nearest"),
    ("fp4_fp8e4m3_bs64_axm1", 8, 8, "fp4_e2m1", "fp8_e4m3", 2, [-1], 64, "nearest"),
    ("fp4_fp8e4m3_bs128_ax0", 8, 8, "fp4_e2m1", "fp8_e4m3", 2, [0], 128, "nearest"),
]


def main():
    scratch, quant, mx_ops, elemwise_ops, formats, linear, specs, posit_mod = _import_reference()
    out = {}

    # (1) a1 format table ---------------------------------------------------
    fmt_names = ["int8", "int4", "int2", "fp8_e5m2", "fp8_e4m3", "fp6_e3m2", "fp6_e2m3", "fp4", "fp4_e2m1",
                 "float16", "fp16", "bfloat16", "bf16"]
    table = {n: [float(v) for v in formats._get_format_params(n)] for n in fmt_names}
    with open(os.path.join(HERE, "format_table.json"), "w") as f:
        json.dump(table, f, indent=1)

    with open(os.path.join(HERE, "kat_vectors.json"), "w") as f:
        json.dump(kat_vectors(), f, indent=1)

    # (2) a2 scalar codec sweeps -------------------------------------------
    x = sweep_inputs()
    d = {"x": x}
    for n in ["int8", "int4", "int2", "fp8_e5m2", "fp8_e4m3", "fp6_e3m2", "fp6_e2m3", "fp4", "fp16", "bf16"]:
        ebits, mbits, emax, max_norm, _ = formats._get_format_params(n)
        for rnd in ("nearest", "floor", "even"):
            for sat in (True, False):
                y = elemwise_ops._quantize_elemwise_core(torch.from_numpy(x.copy()), mbits, ebits, max_norm,
                                                         round=rnd, saturate_normals=sat, allow_denorm=True)
                d[f"{n}|{rnd}|sat{int(sat)}|dn1"] = y.numpy()
        y = elemwise_ops._quantize_elemwise_core(torch.from_numpy(x.copy()), mbits, ebits, max_norm,
                                                 round="nearest", saturate_normals=True, allow_denorm=False)
        d[f"{n}|nearest|sat1|dn0"] = y.numpy()
    np.savez_compressed(os.path.join(HERE, "elemwise_sweep.npz"), **d)

    # (3) a3..a7 outlier fake-quant ------------------------------------------
    ts = tensors()
    d = {}
    meta = {}
    for tname, A in ts.items():
        d[f"in|{tname}"] = A.numpy()
    for (cname, isb, osb, ifmt, ofmt, sd, axes, bs, rnd) in CFGS:
        for tname, A in ts.items():
            if A.ndim == 3 and axes == [0]:
                ax = [1]      # the MXLinear activation convention (linear.py:66-73)
            else:
                ax = axes
            key = f"{cname}|{tname}"
            try:
                y = quant.quantize_mx_outlier_v1(A.clone(), isb, osb, ifmt, ofmt, "max", sd, ax, bs, rnd, False, False)
            except AssertionError as e:
                meta[key] = {"assert": str(e)}
                continue
            # intermediates via the reference's own helpers
            Ab, ax2, orig_shape, padded_shape = quant._reshape_to_blocks(A.clone(), [a % A.ndim for a in ax], bs)
            sea = [a + 1 for a in ax2]
            m = quant._extract_outlier_indices(Ab, sd, sea)
            m_un = quant._undo_reshape_to_blocks(m.clone(), padded_shape, orig_shape, ax2)
            d[f"out|{key}"] = y.numpy()
            d[f"mask|{key}"] = m_un.numpy().astype(np.uint8)
            meta[key] = {"cfg": [isb, osb, ifmt, ofmt, sd, ax, bs, rnd], "density": float(m_un.mean())}
    # a7 hessian variant on columns (llm/gptq.py:130-131)
    for tname in ("gauss", "heavy", "ragged"):
        col = ts[tname][:, 5:6].contiguous()
        q, n_out = quant.quantize_mx_outlier_hessian(col.clone(), 8, 8, "int2", "fp4", "max", 2, [0], 16, "nearest", False, False, False)
        d[f"hess_out|{tname}"] = q.numpy()
        d[f"hess_nout|{tname}"] = n_out.numpy()
    # fp16 / bf16 inputs (RTN path quantizes in the checkpoint dtype, llm/llama.py:238): recorded
    # for the documented dtype delta, compared with a tolerance not bit-exactly.
    for dt, nm in ((torch.float16, "f16"), (torch.bfloat16, "bf16")):
        A = ts["fp16rep"].to(dt)
        y = quant.quantize_mx_outlier_v1(A.clone(), 8, 8, "fp4_e2m1", "fp8_e4m3", "max", 2, [-1], 32, "nearest", False, False)
        d[f"lowp_in|{nm}"] = A.float().numpy()
        d[f"lowp_out|{nm}"] = y.float().numpy()
    np.savez_compressed(os.path.join(HERE, "outlier_fakequant.npz"), **d)
    with open(os.path.join(HERE, "outlier_fakequant_meta.json"), "w") as f:
        json.dump(meta, f, indent=1)

    # (4) a10 mx_ops variant (used by MXLinear) ------------------------------
    d = {}
    g = torch.Generator().manual_seed(99)
    W = torch.randn(96, 128, generator=g) * 0.05
    X = torch.randn(64, 128, generator=g)
    d["W"] = W.numpy(); d["X"] = X.numpy()
    for nm, A in (("W", W), ("X", X)):
        for fmt, sb in (("fp6_e3m2", 4), ("fp4_e2m1", 8), ("fp8_e4m3", 8)):
            y = mx_ops._quantize_mx_outlier_v1(A.clone(), sb, sb, fmt, fmt, "max", 5, [1], 32, "nearest", False, False)
            d[f"v1|{nm}|{fmt}|sb{sb}"] = y.numpy()
    np.savez_compressed(os.path.join(HERE, "mxops_variant.npz"), **d)

    # (5) a9 _quantize_mx: seeded tensors, both with and without the +1e-6 defect
    d = {}
    for tname in ("gauss", "heavy", "ragged", "big", "act3d"):
        A = ts[tname]
        for fmt in ("fp8_e4m3", "fp6_e3m2", "fp4_e2m1", "int8", "int4"):
            for (axes, bs) in (([-1], 32), ([0], 16)):
                y_def = mx_ops._quantize_mx(A.clone(), 8, fmt, "max", axes, bs, "nearest", False, False)
                d[f"defect|{tname}|{fmt}|ax{axes[0]}|bs{bs}"] = y_def.numpy()
    # upstream semantics: same function with the `+ 1e-6` removed, obtained by
    # running the reference function on a module copy whose constant is zeroed
    import inspect, types
    src = inspect.getsource(mx_ops._quantize_mx).replace("+ 1e-6", "+ 0.0")
    ns = dict(mx_ops.__dict__)
    exec(compile(src, "<reference _quantize_mx without +1e-6>", "exec"), ns)
    qmx_up = ns["_quantize_mx"]
    for tname in ("gauss", "heavy", "ragged", "big", "act3d"):
        A = ts[tname]
        for fmt in ("fp8_e4m3", "fp6_e3m2", "fp4_e2m1", "int8", "int4"):
            for (axes, bs) in (([-1], 32), ([0], 16)):
                y = qmx_up(A.clone(), 8, fmt, "max", axes, bs, "nearest", False, False)
                d[f"upstream|{tname}|{fmt}|ax{axes[0]}|bs{bs}"] = y.numpy()
    for tname in ("gauss", "heavy", "ragged", "big", "act3d"):
        d[f"in|{tname}"] = ts[tname].numpy()
    np.savez_compressed(os.path.join(HERE, "quantize_mx.npz"), **d)

    # (6) a11 MXLinear forward ----------------------------------------------
    d = {}
    g = torch.Generator().manual_seed(7)
    Xl = torch.randn(64, 128, generator=g)
    Wl = torch.randn(512, 128, generator=g) * 0.05
    bl = torch.randn(512, generator=g) * 0.1
    d["X"] = Xl.numpy(); d["W"] = Wl.numpy(); d["b"] = bl.numpy()
    spec_sets = {
        "fp6": {"w_elem_format": "fp6_e3m2", "a_elem_format": "fp6_e3m2", "scale_bits": 4, "block_size": 32, "bfloat": 16, "custom_cuda": False},
        "w4a8": {"w_elem_format": "fp4_e2m1", "a_elem_format": "fp8_e4m3", "scale_bits": 8, "block_size": 32, "bfloat": 16, "custom_cuda": False},
        "w4a8_nobf": {"w_elem_format": "fp4_e2m1", "a_elem_format": "fp8_e4m3", "scale_bits": 8, "block_size": 32, "custom_cuda": False},
    }
    for nm, sp in spec_sets.items():
        sp = specs.finalize_mx_specs(dict(sp))
        lin = linear.MXLinear(128, 512, True, mx_specs=sp)
        with torch.no_grad():
            lin.weight.copy_(Wl); lin.bias.copy_(bl)
            y = lin(Xl)
        d[f"y|{nm}"] = y.numpy()
    np.savez_compressed(os.path.join(HERE, "mxlinear.npz"), **d)

    # (7) a13 posit tables -----------------------------------------------------
    d = {}
    Posit = posit_mod.Posit
    for (n, es) in ((8, 1), (8, 0), (8, 2), (6, 1), (4, 1)):
        vals = []
        for code in range(2 ** n):
            p = Posit(0, n, es)
            p.set_bit_pattern(code)
            v = p.get_value()
            vals.append(float(v) if code != 2 ** (n - 1) else float("nan"))
        d[f"decode|{n}|{es}"] = np.array(vals, dtype=np.float64)
        rng = np.random.RandomState(n * 10 + es)
        xs = np.concatenate([rng.randn(600) * np.exp2(rng.randint(-14, 14, 600)),
                             np.array(vals)[np.isfinite(vals)],
                             # midpoints between adjacent posits (ties)
                             ]).astype(np.float64)
        fin = np.sort(np.array([v for v in vals if np.isfinite(v)]))
        mids = (fin[:-1] + fin[1:]) / 2.0
        xs = np.concatenate([xs, mids, mids * (1 + 1e-6), mids * (1 - 1e-6)])
        xs = xs.astype(np.float32).astype(np.float64)
        codes = np.array([Posit(float(v), n, es).number for v in xs], dtype=np.int64)
        d[f"enc_x|{n}|{es}"] = xs
        d[f"enc_code|{n}|{es}"] = codes
    np.savez_compressed(os.path.join(HERE, "posit.npz"), **d)


    # (8) a14 tiny-model perplexity (llm/llama.py:176-284 formula; weights quantised by the reference's
    # quantize_mx_outlier_v1 with the harness configuration llm/llama.py:229-237, whole-model forward) ----
    import torch.nn as nn
    from transformers import LlamaConfig, LlamaForCausalLM, OPTConfig, OPTForCausalLM
    d = {}
    def ppl_of(model, ids, seqlen):
        n = ids.numel() // seqlen
        nlls = []
        with torch.no_grad():
            for i in range(n):
                b = ids[:, i * seqlen:(i + 1) * seqlen]
                lg = model(b).logits
                loss = nn.CrossEntropyLoss()(lg[:, :-1, :].reshape(-1, lg.size(-1)), b[:, 1:].reshape(-1))
                nlls.append(loss.float() * seqlen)
        return torch.exp(torch.stack(nlls).sum() / (n * seqlen)).item()
    def find_linears(mod, name=''):
        if type(mod) is nn.Linear:
            return {name: mod}
        r = {}
        for n1, c in mod.named_children():
            r.update(find_linears(c, name + '.' + n1 if name else n1))
        return r
    torch.manual_seed(0)
    models = {
        "llama": (LlamaForCausalLM(LlamaConfig(hidden_size=64, intermediate_size=176, num_hidden_layers=2,
                                               num_attention_heads=4, num_key_value_heads=4, vocab_size=128,
                                               max_position_embeddings=64)), lambda m: m.model.layers),
        "opt": (OPTForCausalLM(OPTConfig(hidden_size=64, ffn_dim=256, num_hidden_layers=2, num_attention_heads=4,
                                         vocab_size=128, max_position_embeddings=64, word_embed_proj_dim=64)),
                lambda m: m.model.decoder.layers),
    }
    g = torch.Generator().manual_seed(11)
    ids = torch.randint(0, 128, (1, 32 * 8 + 5), generator=g)
    d["tokens"] = ids.numpy()
    for nm, (model, layers_of) in models.items():
        model.eval()
        with torch.no_grad():      # give the weights a realistic spread incl. a heavy tail
            for p_ in model.parameters():
                if p_.ndim == 2:
                    p_.mul_(1.0 + 4.0 * (torch.rand(p_.shape, generator=g) < 0.02).float())
        for k, v in model.state_dict().items():
            d[f"{nm}|sd|{k}"] = v.numpy()
        d[f"{nm}|ppl_fp32"] = np.float64(ppl_of(model, ids, 32))
        for cfgname, (fi, fo, ax, bs) in {"int2_fp4_ax0": ("int2", "fp4", [0], 16),
                                          "fp4_fp8_axm1": ("fp4_e2m1", "fp8_e4m3", [-1], 32)}.items():
            import copy
            mq = copy.deepcopy(model)
            for layer in layers_of(mq):
                for lname, lin in find_linears(layer).items():
                    lin.weight.data = quant.quantize_mx_outlier_v1(lin.weight.data, 8, 8, fi, fo, "max", 2, ax, bs,
                                                                   "nearest", False, False)
            d[f"{nm}|{cfgname}|ppl"] = np.float64(ppl_of(mq, ids, 32))
            l0 = find_linears(layers_of(mq)[0])
            for lname in sorted(l0)[:2]:
                d[f"{nm}|{cfgname}|w|{lname}"] = l0[lname].weight.data.numpy()
            tot = 0.0
            for layer in layers_of(mq):
                for lname, lin in find_linears(layer).items():
                    tot += float(lin.weight.data.double().abs().sum())
            d[f"{nm}|{cfgname}|abs_sum"] = np.float64(tot)
    np.savez_compressed(os.path.join(HERE, "tiny_ppl.npz"), **d)
    print({k: float(v) for k, v in d.items() if k.endswith("ppl") or k.endswith("ppl_fp32")})


    # (10) f1 GPTQ + MicroScopiQ pruning (llm/gptq.py:60-184) on a small Linear, reference solver on the CPU
    import types
    gsrc = open(os.path.join(REF, "llm", "gptq.py")).read()
    gmod = types.ModuleType("ref_gptq")
    exec(compile(gsrc.replace("torch.cuda.synchronize()", "pass"), "<reference llm/gptq.py, cuda sync removed>", "exec"), gmod.__dict__)
    d = {}
    g = torch.Generator().manual_seed(21)
    lin = torch.nn.Linear(48, 64, bias=False)
    with torch.no_grad():
        lin.weight.copy_(torch.randn(64, 48, generator=g) * 0.05)
    X = torch.randn(4, 32, 48, generator=g)
    d["W"] = lin.weight.detach().numpy().copy(); d["X"] = X.numpy()
    gp = gmod.GPTQ(lin)
    gp.quantizer = quant.MXQuantizer()
    gp.quantizer.configure(8, 8, "int2", "fp4", axes=[0], block_size=16)
    for b in range(4):
        gp.add_batch(X[b], None)
    d["H"] = gp.H.numpy().copy()
    import io, contextlib
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        gp.fasterquant(blocksize=16, percdamp=.01)
    d["Q"] = lin.weight.detach().numpy().copy()
    d["error"] = np.float64(float(buf.getvalue().split("error")[1]))
    Wq_rtn = quant.quantize_mx_outlier_v1(torch.from_numpy(d["W"]), 8, 8, "int2", "fp4", "max", 2, [0], 16)
    Y = X.reshape(-1, 48) @ torch.from_numpy(d["W"]).t()
    d["out_err_gptq"] = np.float64(((X.reshape(-1, 48) @ torch.from_numpy(d["Q"]).t() - Y) ** 2).sum().item())
    d["out_err_rtn"] = np.float64(((X.reshape(-1, 48) @ Wq_rtn.t() - Y) ** 2).sum().item())
    np.savez_compressed(os.path.join(HERE, "gptq.npz"), **d)
    print("gptq fixture: error", float(d["error"]), "out err gptq/rtn", float(d["out_err_gptq"]), float(d["out_err_rtn"]))

    # (9) torch CPU reduction-order pins (what torch.mean / torch.std compute) ---
    d = {}
    g = torch.Generator().manual_seed(5)
    for bs in (8, 16, 32, 64, 128):
        for I in (40, 6, 1):     # 40: 32 cascade + 8 ilp4 columns; 6: 4 cascade + 2 ilp4; 1: inner order
            a = (torch.randn(12, bs, I, generator=g).abs() * 0.02)
            d[f"outer_in|{bs}|{I}"] = a.numpy()
            d[f"outer_mean|{bs}|{I}"] = torch.mean(a, dim=[1], keepdim=True).numpy()
            d[f"outer_std|{bs}|{I}"] = torch.std(a, dim=[1], keepdim=True, unbiased=False).numpy()
        b = (torch.randn(40, 3, bs, generator=g).abs() * 0.02)
        d[f"inner_in|{bs}"] = b.numpy()
        d[f"inner_mean|{bs}"] = torch.mean(b, dim=[2], keepdim=True).numpy()
        d[f"inner_std|{bs}"] = torch.std(b, dim=[2], keepdim=True, unbiased=False).numpy()
    np.savez_compressed(os.path.join(HERE, "reduce_order.npz"), **d)

    shutil.rmtree(scratch, ignore_errors=True)
    for fn in sorted(os.listdir(HERE)):
        print(fn, os.path.getsize(os.path.join(HERE, fn)))


if __name__ == "__main__":
    main()
