"""SURVEY 8 row a2 (`_quantize_elemwise_core` / `_round_mantissa`, number_system/mx/elemwise_ops.py:47-174): the ONE magnitude per grid on which
the reference's `nearest` / `even` are not what their names say.  `floor(|x| + 0.5)` is computed in float32 (elemwise_ops.py:64-65); for
|x| = pred(0.5) on the integer grid the sum 1 - 2^-25 is a tie of float32 and rounds to 1.0, so the reference returns the grid's smallest step
where true rounding returns 0.  Unscaled: |x| = pred(h), h = half the smallest subnormal of the element format times the block scale.  Found in
round 6 (a planted pred(2^5) beside a block maximum of 713 in the strided-axis `_quantize_mx` test).  The oracle computes the sum in float as the
reference does; every bit / hardware-convert codec of the library has to special-case it.  Every surface below is driven with planted quirk
values (and with their float neighbours, which must NOT be touched) and compared with the oracle bit for bit."""
import numpy as np
import pytest
import torch

from gpu_common import dev, eq

pytestmark = pytest.mark.gpu


def _pred(x, n=1):
    return np.uint32(np.float32(x).view(np.uint32) - np.uint32(n)).view(np.float32)


def _succ(x):
    return np.uint32(np.float32(x).view(np.uint32) + np.uint32(1)).view(np.float32)


def _plant_rows(K, block, half_step_exp_of, seed):
    """rows of `K` floats in blocks of `block`: per block a maximum `top` (sets the scale), then pred(h), -pred(h), h, pred(pred(h)), succ(h)
    with h = 2^half_step_exp_of(top), the rest small noise"""
    rs = np.random.RandomState(seed)
    rows = []
    for u in range(-20, 21):
        r = (rs.randn(K) * 1e-3).astype(np.float32)
        for b in range(K // block):
            top = np.float32((1.0 + 0.75 * rs.rand()) * 2.0 ** (u + b % 3))
            h = np.float32(2.0 ** half_step_exp_of(int(np.floor(np.log2(top)))))
            blk = r[b * block:(b + 1) * block]
            blk *= np.float32(h * 4)
            blk[0] = top if b % 2 else -top
            blk[1:6] = [_pred(h), -_pred(h), h, _pred(h, 2), _succ(h)]
        rows.append(r)
    return np.stack(rows)


@pytest.mark.parametrize("fmt,emax,half", [("fp4_e2m1", 2, -2), ("fp8_e4m3", 8, -10), ("fp6_e3m2", 4, -5), ("fp6_e2m3", 2, -4), ("int4", 0, -3), ("int8", 0, -7)])
@pytest.mark.parametrize("rnd", ["nearest", "even"])
def test_quantize_mx_python_path_has_the_quirk_native_path_has_not(msq, O, fmt, emax, half, rnd):
    A = _plant_rows(256, 32, lambda e: e - emax + half, 3)
    At = torch.from_numpy(A).to(dev())
    y = msq.mx_ops._quantize_mx(At, 8, fmt, axes=[-1], block_size=32, round=rnd).cpu().numpy()
    yo = O.quantize_mx(A, 8, fmt, axis=-1, block_size=32, round=rnd)
    assert eq(y, yo)
    assert (yo[:, 1] != 0).all() and (yo[:, 4] == 0).all()                       # pred(h) -> one step; pred(pred(h)) -> 0: the fixture bites
    yn = msq.mx_ops._quantize_mx(At, 8, fmt, axes=[-1], block_size=32, round=rnd, custom_cuda=True).cpu().numpy()
    e, m, _, mx, _ = msq.formats._get_format_params(fmt)
    assert eq(yn, O.quantize_mx_native(A, 8, e, m, mx, 32, 1, False, {"nearest": 0, "even": 2}[rnd]))
    assert (yn[:, 1] == 0).all()                                                 # integer rounding: 0
    B = np.ascontiguousarray(A.reshape(-1, 8, 32).transpose(0, 2, 1))            # strided-axis kernels
    yb = msq.mx_ops._quantize_mx(torch.from_numpy(B).to(dev()), 8, fmt, axes=[1], block_size=32, round=rnd).cpu().numpy()
    assert eq(yb, O.quantize_mx(B, 8, fmt, axis=1, block_size=32, round=rnd))


@pytest.mark.parametrize("fi,fo", [("fp4_e2m1", "fp8_e4m3"), ("fp4_e2m1", "posit8_es1"), ("int2", "fp4"), ("fp4_e2m1", "fp4_e2m1"), ("fp8_e4m3", "fp8_e4m3")])
@pytest.mark.parametrize("axis,bs", [(-1, 32), (0, 16)])
def test_outlier_fakequant_fast_paths_fall_back_on_the_quirk(msq, O, fi, fo, axis, bs):
    """a6 (utils/quant.py:147-266) through the hardware-convert paths: a block that holds an all-ones-mantissa element takes the arithmetic codec"""
    emax = {"fp4_e2m1": 2, "int2": 0, "fp8_e4m3": 8}[fi]
    half = {"fp4_e2m1": -2, "int2": -1, "fp8_e4m3": -10}[fi]
    A = _plant_rows(256, bs, lambda e: e - emax + half, 5)
    # the planted maximum is an outlier in most blocks (it sets e_out, not e_in): add rows where the block is flat enough to have none
    rs = np.random.RandomState(9)
    flat = (1.0 + 0.2 * rs.rand(*A.shape)).astype(np.float32) * np.float32(0.75)
    h = np.float32(2.0 ** (-1 - emax + half))                                    # max in [0.75, 0.9] -> exponent -1
    flat[:, 1::bs] = _pred(h); flat[:, 2::bs] = -_pred(h); flat[:, 3::bs] = h
    A = np.concatenate([A, flat.astype(np.float32)])
    if axis == 0:
        A = np.ascontiguousarray(A.reshape(A.shape[0], -1, bs).transpose(2, 0, 1).reshape(bs, -1))    # blocks down the columns
        A = np.ascontiguousarray(np.tile(A, (2, 1)))
    At = torch.from_numpy(A).to(dev())
    r = msq.quant.outlier_fakequant(At, 8, 8, fi, fo, 2, axis, bs, want_mask=True)
    o = O.outlier_fakequant(A, 8, 8, fi, fo, 2, axis, bs)
    assert (r["mask"].cpu().numpy() == o["mask"]).all()
    assert eq(r["out"].cpu().numpy(), o["out"])


def test_mx_operand_packers_have_the_quirk(msq, O):
    """msq_mx_pack_a8 (both kernels), the fused RMSNorm / silu packers share mx_pack8_e4m3_quad; msq_mx_pack_w4 / _w6 through the GEMM"""
    X = _plant_rows(512, 32, lambda e: e - 8 - 10, 11)
    Xo = O.quantize_mx(X, 8, "fp8_e4m3", axis=-1, block_size=32)
    xc, xs = msq.qlinear.mx_pack_act(torch.from_numpy(X).to(dev()), check_status=True)
    c = xc.cpu().numpy().astype(np.int32)
    dec = np.where(c & 0x80, -1.0, 1.0) * np.where((c >> 3) & 15, (1 + (c & 7) / 8.0) * np.exp2(((c >> 3) & 15) - 7.0), (c & 7) / 8.0 * 2.0 ** -6)
    dec = dec * np.repeat(np.exp2(xs.cpu().numpy().astype(np.float64) - 127.0), 32, axis=1)
    assert (dec == Xo.astype(np.float64)).all()
    assert (Xo[:, 1] != 0).all()
    eye = torch.eye(256, device=dev())
    for w_fmt, ofmt, emax, half in (("e2m1", "fp4_e2m1", 2, -2), ("e3m2", "fp6_e3m2", 4, -5), ("e2m3", "fp6_e2m3", 2, -4)):
        W = _plant_rows(256, 32, lambda e: e - emax + half, 13)
        W = np.concatenate([W] * 7)[:256]
        P = msq.qlinear.mx_pack_weight(torch.from_numpy(W).to(dev()), w_fmt=w_fmt)
        Wd = msq.qlinear.qlinear_mx_w4a8(eye, P, None, torch.float32).t().cpu().numpy()
        Wo = O.quantize_mx(W, 8, ofmt, axis=-1, block_size=32)
        assert np.array_equal(Wd, Wo), w_fmt
        assert (Wo[:, 1] != 0).all()


def test_activation_quantiser_falls_back_on_the_quirk(msq, O):
    """msq_act_quant (a10 on the activation, e4m3 / e4m3, std_dev 2 here): act_block_lean returns false on an all-ones mantissa"""
    X = _plant_rows(256, 32, lambda e: e - 8 - 10, 17)
    xq, st = msq.qlinear.act_quant(torch.from_numpy(X).to(dev()), 8, 8, "fp8_e4m3", "fp8_e4m3", 2, 32)
    xo = O.outlier_fakequant(X, 8, 8, "fp8_e4m3", "fp8_e4m3", 2, -1, 32)["out"]
    assert (xq.float().cpu().numpy() == xo).all()


def test_reference_made_quirk_fixture(msq):
    """the same planted values through the REFERENCE (tests/golden/half_away_quirk.npz, make_golden_quirk.py): a6 bit for bit, `_quantize_mx`
    under reference_python_divisor() (= the reference's CPU arithmetic, `+1e-6` included) bit for bit"""
    import os
    from gpu_common import G
    z = np.load(os.path.join(G, "half_away_quirk.npz"))
    n = 0
    for k in z.files:
        if k.startswith("mx|"):
            _, fmt, rnd = k.split("|")
            with msq.mx_ops.reference_python_divisor():
                y = msq.mx_ops._quantize_mx(torch.from_numpy(z["in|mx|" + fmt]).to(dev()), 8, fmt, axes=[-1], block_size=32, round=rnd)
            assert eq(y.cpu().numpy(), z[k]), k
            n += 1
        elif k.startswith("a6|"):
            _, fi, fo, ax, bs = k.split("|")
            y = msq.quant.quantize_mx_outlier_v1(torch.from_numpy(z["in|" + k]).to(dev()), 8, 8, fi, fo, "max", 2, [int(ax[2:])], int(bs[2:]), "nearest")
            assert eq(y.cpu().numpy(), z[k]), k
            n += 1
    assert n == 16
